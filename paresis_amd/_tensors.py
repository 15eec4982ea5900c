"""Host <-> HBM plumbing shared by the class mirrors: accept numpy or torch, hand HIP a contiguous cuda tensor."""
import numpy as np
import torch

from ._lib import PsxError


def device():
    if not torch.cuda.is_available():
        raise PsxError("no MI355X visible (torch.cuda.is_available() is False): the hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(a, dtype):
    """numpy / torch (any device) -> contiguous tensor of `dtype` in HBM."""
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype != dtype or not t.is_cuda:
        t = t.to(device=device(), dtype=dtype)
    return t.contiguous()


def is_scalar(x):
    return isinstance(x, (int, float)) or (isinstance(x, np.ndarray) and x.ndim == 0)
