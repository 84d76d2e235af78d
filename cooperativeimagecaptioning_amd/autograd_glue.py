"""The single autograd node of a training step.

The HIP engines compute their own backward pass; torch autograd only has to call it when the
caller runs ``loss.backward()`` (train.py:203-208).  EngineLoss returns the engine-computed
loss value and, on backward, invokes a closure that launches the backward engines, which
accumulate straight into the flat gradient buffers the parameters' ``.grad`` views alias.
"""
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
