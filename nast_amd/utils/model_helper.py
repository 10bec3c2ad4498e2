"""BaseModel: the slice of utils/model_helper.py:17-101 the training step depends on (counters, device
placement).  Checkpoint rotation / annealing helpers are out of scope."""
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
        self.net = self.net.to(args.cuda)

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
