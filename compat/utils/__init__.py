"""`from utils import ...` (styletransformer/seq2seq.py:14, main.py:8) -> nast_amd.utils."""
import os as _os

from nast_amd.utils import *          # noqa: F401,F403
from nast_amd.utils import __all__ as _names
from . import gumbel                 # noqa: F401   (`utils.gumbel`)

# sub-packages that are not part of the hot path (utils.cotk_private: cotk data loading) come from the reference checkout
_ref = _os.environ.get("NAST_REFERENCE_ROOT")
if _ref and _os.path.isdir(_os.path.join(_ref, "utils")):
    __path__.append(_os.path.join(_ref, "utils"))


def debug():
    """utils/debug_helper.py attaches a remote debugger; nothing to attach here."""


def try_cache(module, args, f_args=None, name=None):
    """utils/cache_helper.py memoises data loading on disk; here: call through."""
    return module(*args) if f_args is None else module(*args, **f_args)


__all__ = list(_names) + ["debug", "try_cache", "gumbel"]
