"""Persistent device buffers keyed by role and shape: outputs, noise and workspaces of the engines are written into
the same tensors every step (no allocator traffic inside the step)."""
import torch


class BufCache:
    def __init__(self):
        self._b = {}

    def get(self, key, shape, dtype, device, fill=None):
        k = (key, tuple(shape), dtype, str(device))
        t = self._b.get(k)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            if fill is not None:
                t.fill_(fill)
            self._b[k] = t
        return t

    def stage(self, key, src, dtype=None):
        """`src` as a contiguous tensor of `dtype`: used in place when it already is one (no copy, no launch), else
        converted into the persistent buffer of its role."""
        dtype = dtype or src.dtype
        if src.dtype == dtype and src.is_contiguous():
            return src
        t = self.get(key, src.shape, dtype, src.device)
        t.copy_(src)
        return t
