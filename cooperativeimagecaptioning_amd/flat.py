"""Flat parameter / gradient / Adam-state buffers for one agent (speaker or listener).

MI355X-first layout: all parameters of an agent live in ONE contiguous f32 buffer (the
nn.Parameters are views into it, so state-dict names and shapes stay those of the
reference), and so do their gradients.  Consequences: the clamp+Adam update is one kernel
launch per agent and the data-parallel gradient exchange is one RCCL all-reduce per agent
 instead of one per tensor.
"""
import torch


class FlatAgent:
    ALIGN = 64  # floats: every parameter starts on a 256-byte boundary (float4 / MFMA tile loads)

    def __init__(self, module, tail=(), external=()):
        """tail: parameter names laid out LAST in the flat buffers, in that order (state-dict order is untouched).
        The speaker puts its logit layer there: its gradient is final before the BPTT loop of the backward pass starts,
        so [tail_offset, numel) is a contiguous bucket whose all-reduce can travel under the rest of backward.
        external: parameter names that live in ANOTHER agent's flat buffers (share_embed = 1: the one embedding table of
        AlternatingJointModel.py:83-88 is the listener's parameter; the speaker reads it, adds its gradient into the listener's
        gradient segment and steps it with moments of its own - optimizer.FlatAdam).  They get no offset here; tensors() /
        grad_tensors() hand out whatever their owner points them at."""
        self.module = module
        self.params = [p for p in module.parameters()]
        self.names = [n for n, _ in module.named_parameters()]
        self.external = [n for n in external if n in self.names]
        tail = [n for n in tail if n in self.names]
        order = [i for i, n in enumerate(self.names) if n not in tail and n not in self.external] + \
            [self.names.index(n) for n in tail]
        self.offsets = [None if n in self.external else 0 for n in self.names]
        off = 0
        self.tail_offset = None
        for i in order:
            if tail and self.names[i] == tail[0]:
                self.tail_offset = off
            self.offsets[i] = off
            off += (self.params[i].numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        if self.tail_offset is None:
            self.tail_offset = off
        self.flat = None
        self.grad = None
        self.exp_avg = None
        self.exp_avg_sq = None
        self.step = 0
        # True from the moment the gradient buffer is handed out for writing (grad_tensors(), a fresh attach()) until a fused
        # clamp+Adam+zero step has cleared it: FlatAdam.zero_grad() may skip its fill only while this is False
        self.grad_dirty = True

    def attached(self):
        if self.flat is None:
            return False
        return all(p.data_ptr() == self.flat.data_ptr() + 4 * o for p, o in zip(self.params, self.offsets) if o is not None)

    def attach(self):
        """(Re-)create the flat buffers on the parameters' current device and re-point the views."""
        dev = next(p for p, o in zip(self.params, self.offsets) if o is not None).device
        flat = torch.zeros(self.numel, device=dev)
        grad = torch.zeros(self.numel, device=dev)
        for p, o in zip(self.params, self.offsets):
            if o is None:
                continue
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
            g_old = p.grad
            p.grad = grad[o:o + p.numel()].view(p.shape)
            if g_old is not None:
                p.grad.copy_(g_old)
        self.flat, self.grad = flat, grad
        self.grad_dirty = True
        if self.exp_avg is None or self.exp_avg.device != dev:
            self.exp_avg = torch.zeros(self.numel, device=dev)
            self.exp_avg_sq = torch.zeros(self.numel, device=dev)

    def ensure(self):
        if not self.attached():
            self.attach()
        else:
            # a caller may have replaced p.grad (zero_grad(set_to_none=True)); re-point it
            for p, o in zip(self.params, self.offsets):
                if o is not None and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
                    p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def tensors(self, prefix=''):
        return {prefix + n: p.data for n, p in zip(self.names, self.params)}

    def grad_tensors(self, prefix=''):
        self.grad_dirty = True            # whoever asks for these writes into them (the backward engines)
        if self.external and getattr(self, 'ext_owner_flat', None) is not None:
            self.ext_owner_flat.grad_dirty = True     # ... and into the owner's gradient segment of a shared table
        # an external parameter's gradient is its owner's segment (p.grad, set by the owner's ensure()); a FROZEN one gets
        # none (the engines skip a null pointer): the reference's autograd would not touch it either
        return {prefix + n: (self.grad[o:o + p.numel()].view(p.shape) if o is not None else (p.grad if p.requires_grad else None))
                for n, p, o in zip(self.names, self.params, self.offsets)}

    def zero_grad(self):
        self.ensure()
        self.grad.zero_()
