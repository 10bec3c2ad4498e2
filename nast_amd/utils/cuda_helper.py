"""Device shim with the reference's call signatures (utils/cuda_helper.py:40-82), device-agnostic inside:
`init(device, "cuda"|"cpu")` selects where `Tensor/LongTensor/zeros/ones/cuda` allocate."""
import torch

_device = torch.device("cpu")


def init(device, kind):
    global _device
    if kind == "cuda":
        _device = torch.device("cuda", device)
        torch.cuda.set_device(_device)
    elif kind == "cpu":
        _device = torch.device("cpu")
    else:
        raise RuntimeError("Please use cuda or cpu for args.cuda")


def _resolve(device):
    if device is None:
        return _device
    if isinstance(device, torch.Tensor):
        return device.device
    if isinstance(device, int):
        return torch.device("cuda", device)
    return torch.device(device)


def Tensor(*x, device=None):
    if len(x) == 1 and not isinstance(x[0], int):
        return torch.as_tensor(x[0], dtype=torch.float32).to(_resolve(device))
    return torch.empty(*x, dtype=torch.float32, device=_resolve(device))


def LongTensor(*x, device=None):
    if len(x) == 1 and not isinstance(x[0], int):
        return torch.as_tensor(x[0]).to(torch.int64).to(_resolve(device))
    return torch.empty(*x, dtype=torch.int64, device=_resolve(device))


def cuda(x, device=None):
    return x.to(_resolve(device))


def zeros(*size, device=None):
    return torch.zeros(*size, dtype=torch.float32, device=_resolve(device))


def ones(*size, device=None):
    return torch.ones(*size, dtype=torch.float32, device=_resolve(device))
