"""BaseModel: the slice of utils/model_helper.py:17-101 the training step depends on (counters, device
placement, `args.restore` of a checkpoint FILE).  Checkpoint rotation (the manager's "last" / "best" names), URL
downloads and annealing helpers are out of scope."""
import os

import torch


class BaseModel:
    def __init__(self, param, net, optimizerList, checkpoint_manager=None, helperList=None):
        self.param = param
        self.args = args = param.args
        self.net = net
        self.optimizerList = optimizerList
        self.now_batch = 0
        self.now_epoch = 0
        self.checkpoint_manager = checkpoint_manager
        if args.cuda is True:
            raise RuntimeError("Please use cuda or cpu for args.cuda")
        restored = args.get("restore") is not None
        if restored:
            self.restore(args.restore)                       # model_helper.py:53-88
            cb = param.volatile.get("restoreCallback")
            if cb:
                cb(self)                                     # model_helper.py:98-99
        self.net = self.net.to(args.cuda)

    def restore(self, path):
        """`args.restore` (model_helper.py:53-88) for a checkpoint file: counters, weights, and -- with
        `args.restore_optimizer` -- the optimizer states, from the dictionary `save_checkpoint` writes (the reference's layout:
        now_batch / now_epoch / weights / one entry per optimizer).  The file is read with `weights_only=True` (nothing in it
        is executed); a name the reference's checkpoint manager would resolve ("last", "best", a URL) is not a file here
        and raises instead of being silently ignored."""
        if str(path).startswith("http") or not os.path.isfile(path):
            raise NotImplementedError(f"args.restore={path!r}: only a checkpoint file path is supported on this path "
                                      "(the checkpoint manager's names and URLs are out of scope, SURVEY 2.1 row 11)")
        ck = torch.load(path, map_location="cpu", weights_only=True)
        if self.args.get("restore_other_weights"):
            raise NotImplementedError("args.restore_other_weights: annealed hyper-parameters (other_weights) are out of scope "
                                      "(SURVEY 2.1 row 11)")
        self.now_batch, self.now_epoch = ck["now_batch"], ck["now_epoch"]
        self.net.load_state_dict(ck["weights"], self.param.volatile.get("load_exclude_set"))             # model_helper.py:65 (excluded keys keep their init)
        if self.args.get("restore_optimizer", True):
            for name, opt in self.optimizerList.items():
                if name in ck and ck[name].get("state"):
                    keep = [{k: v for k, v in g.items() if k != "params"} for g in opt.param_groups]
                    opt.load_state_dict(ck[name])
                    for g, new in zip(opt.param_groups, keep):        # the run's own hyper-parameters win (:70-77)
                        g.update(new)
                    self.optimizerCuda(opt)
        return ck

    def optimizerCuda(self, optimizer):
        """model_helper.py:135-139: load_state_dict casts the moments to the parameters' device AT THAT MOMENT -- the host,
        since the network moves to args.cuda only afterwards and `.to()` moves parameter data, not optimizer state.  The step
        counts stay host scalars (torch.optim.Adam's layout; FusedAdam reads them on the host)."""
        for state in optimizer.state.values():
            for k, v in state.items():
                if torch.is_tensor(v) and k != "step":
                    state[k] = v.to(self.args.cuda)

    def zero_grad(self):
        for p in self.net.parameters():
            if p.grad is not None:
                p.grad.detach_()
                p.grad.zero_()

    def state_dict(self):
        state = {"now_epoch": self.now_epoch, "now_batch": self.now_batch, "args": dict(self.param.args),
                 "weights": self.net.state_dict()}
        for name, opt in self.optimizerList.items():
            state[name] = opt.state_dict()
        return state

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)
