"""Mirror of the reference's misc/utils.py for the hot path (same names and argument meaning).

Reference: /misc/utils.py:15-107.  skimage / scipy.misc imports of the reference are unused
there and dropped here.
"""
import numpy as np
import torch
import torch.nn as nn


def if_use_att(opt):
    """misc/utils.py:15-20."""
    if opt.caption_model in ['show_tell', 'all_img', 'fc'] and opt.vse_model in ['fc', 'fc2']:
        return False
    return True


def decode_sequence(ix_to_word, seq):
    """misc/utils.py:23-37: 0 is the END token."""
    N, D = seq.size()
    out = []
    for i in range(N):
        txt = ''
        for j in range(D):
            ix = int(seq[i, j])
            if ix > 0:
                if j >= 1:
                    txt = txt + ' '
                txt = txt + ix_to_word[str(ix)]
            else:
                break
        out.append(txt)
    return out


def to_contiguous(tensor):
    return tensor if tensor.is_contiguous() else tensor.contiguous()


class LanguageModelCriterion(nn.Module):
    """misc/utils.py:45-58.  On the hot path the masked NLL runs inside the HIP engine
    (cic_masked_nll); this module keeps the reference's interface for callers that hold
    log-probabilities as tensors (evaluation code)."""

    def forward(self, input, target, mask):
        target = target[:, :input.size(1)]
        mask = mask[:, :input.size(1)]
        output = -input.gather(2, target.unsqueeze(2)).squeeze(2) * mask
        return torch.sum(output) / torch.sum(mask)


def set_lr(optimizer, lr):
    """misc/utils.py:60-62."""
    for group in optimizer.param_groups:
        group['lr'] = lr


def clip_gradient(optimizer, grad_clip):
    """misc/utils.py:65-69 — elementwise clamp.  FlatAdam fuses it into its step; for foreign
    optimizers this is the reference behaviour."""
    if hasattr(optimizer, 'set_grad_clip'):
        optimizer.set_grad_clip(grad_clip)
        return
    for group in optimizer.param_groups:
        for param in group['params']:
            if param.grad is not None:
                param.grad.data.clamp_(-grad_clip, grad_clip)


def var_wrapper(x, cuda=True, volatile=False):
    """misc/utils.py:72-87 (Variable/volatile are gone from torch; volatile -> detached)."""
    if type(x) is dict:
        return {k: var_wrapper(v, cuda, volatile) for k, v in x.items()}
    if type(x) is list or type(x) is tuple:
        return [var_wrapper(v, cuda, volatile) for v in x]
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if torch.is_tensor(x):
        x = x.cuda() if cuda else x.cpu()
        if volatile:
            x = x.detach()
    return x


def load_state_dict(model, state_dict):
    """misc/utils.py:89-107: tolerant loader (flatten-and-copy on shape mismatch)."""
    model_state_dict = model.state_dict()
    keys = set(list(model_state_dict.keys()) + list(state_dict.keys()))
    for k in keys:
        if k not in state_dict:
            print(f'key {k} in model.state_dict() not in loaded state_dict')
        elif k not in model_state_dict:
            print(f'key {k} in loaded state_dict not in model.state_dict()')
        else:
            if state_dict[k].size() != model_state_dict[k].size():
                print(f'key {k} size not match in model.state_dict() and loaded state_dict. '
                      f'Try to flatten and copy the values in common parts')
            n = min(model_state_dict[k].numel(), state_dict[k].numel())
            model_state_dict[k].view(-1)[:n].copy_(state_dict[k].view(-1)[:n])
    model.load_state_dict(model_state_dict)
