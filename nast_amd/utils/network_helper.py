"""BaseModule / BaseNetwork: the parameter-group selection by attribute name that splits the F and D
optimisers (utils/network_helper.py:34-47, utils/module_helper.py:4-10)."""
import logging
import re

from torch import nn


class BaseModule(nn.Module):
    def __getattr__(self, key):
        try:
            return super().__getattr__(key)
        except AttributeError:
            for name, value in self.named_children():
                if name.split("_")[0] == key:
                    return value
            raise


class BaseNetwork(BaseModule):
    def __init__(self, param, collection_name=None):
        super().__init__()
        self.args = param.args
        self.volatile = param.volatile
        self.param = param
        self.collection_name = list(collection_name or []) + ["all"]

    def get_parameters_by_name(self, req=None, silent=False):
        """Yield parameters whose qualified name contains `_<req>` not followed by a letter
        (e.g. "F" selects model_F.*), honouring `_exclude_<req>` / `_exclude_all`."""
        for name, p in self.named_parameters():
            if req is None:
                yield p
                continue
            tail = r"(?![a-zA-Z])"
            if re.search("_" + req + tail, name) and not re.search("_exclude_" + req + tail, name) \
                    and not re.search("_exclude_all" + tail, name):
                if not silent:
                    logging.info("\t%s", name)
                yield p

    def load_state_dict(self, state_dict, exclude_set=None, whitelist=None):
        """Tolerant load (utils/network_helper.py:65-121): unexpected / missing keys are logged, not fatal."""
        own = self.state_dict()
        for name, value in state_dict.items():
            if name not in own:
                logging.warning("unexpected key %s in state_dict", name)
                continue
            if exclude_set and name in exclude_set:
                continue
            if whitelist and name not in whitelist:
                continue
            own[name].copy_(value)
        missing = set(own) - set(state_dict)
        if missing:
            logging.warning("missing keys in state_dict: %s", missing)
