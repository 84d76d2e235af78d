"""The single autograd node of a training step.

The HIP engines compute their own backward pass; torch autograd only has to call it when the
caller runs ``loss.backward()`` (train.py:203-208).  EngineLoss returns the engine-computed
loss value and, on backward, invokes a closure that launches the backward engines, which
accumulate straight into the flat gradient buffers the parameters' ``.grad`` views alias.
"""
import weakref

import torch


class EngineLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss_value, anchor, backward_fn):
        # anchor: any tensor that requires grad (a flat parameter buffer) — ties the node into the graph
        ctx.backward_fn = backward_fn
        if getattr(loss_value, '_cic_fresh', False):      # engine.loss_combine: a tensor of this step, not a workspace slot
            return loss_value.detach()
        return loss_value.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        fn, ctx.backward_fn = ctx.backward_fn, None
        if fn is None:
            raise RuntimeError('the engine workspace of this step was already consumed by a backward pass')
        fn(grad_out)
        return None, None, None


_ONES = {}
# True: a bare ``loss.backward()`` calls the step's backward closure directly, on the caller's thread and stream, instead of
# going through torch's autograd engine (whose device thread hand-over showed as a ~25 us hole between the forward and the
# backward launches of every step: profiles/r03_step_sequence.txt).  The autograd node stays in place for every other use.
DIRECT_BACKWARD = [True]


def engine_loss(loss_value, anchor, backward_fn):
    """EngineLoss.apply whose result answers a bare ``loss.backward()`` (train.py:203-208) with a cached tensor of ones as the
    upstream gradient: autograd otherwise makes one with a fill launch (5 us on a ~3 ms step) every iteration.  Any other use of the
    loss (arithmetic on it, ``backward(gradient=...)``, ``torch.autograd.grad``) takes the ordinary path."""
    cell = [backward_fn]

    def run(go):
        fn, cell[0] = cell[0], None
        if fn is None:
            raise RuntimeError('the engine workspace of this step was already consumed by a backward pass')
        fn(go)
    out = EngineLoss.apply(loss_value, anchor, run)
    tensor_backward = torch.Tensor.backward
    # the closure holds the loss WEAKLY: a strong reference would make loss -> closure -> loss a cycle, and the step's decode /
    # listener results (kept alive by the autograd node) would wait for the cyclic collector instead of dying with the loss
    ref = weakref.ref(out)
    key = (out.device, out.dtype, tuple(out.shape))

    def backward(gradient=None, *args, **kwargs):
        me = ref()
        if me is None:
            raise RuntimeError('backward() of a loss tensor that no longer exists')
        if gradient is None:
            gradient = _ONES.get(key)
            if gradient is None:
                gradient = _ONES[key] = torch.ones(key[2], dtype=key[1], device=key[0])
            if DIRECT_BACKWARD[0] and not args and not kwargs:
                with torch.no_grad():             # as inside an autograd node: the closure's own torch ops record nothing
                    run(gradient)
                return None
        return tensor_backward(me, gradient, *args, **kwargs)
    out.backward = backward
    return out
