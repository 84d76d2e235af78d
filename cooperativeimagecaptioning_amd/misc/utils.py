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
    """Token ids -> sentences (the reference's misc/utils.py:23-37 contract: ix_to_word is keyed by the decimal
    string of the id, 0 ends a caption and is not printed)."""
    rows = torch.as_tensor(seq).cpu().tolist()
    sentences = []
    for row in rows:
        end = row.index(0) if 0 in row else len(row)
        sentences.append(' '.join(ix_to_word[str(tok)] for tok in row[:end]))
    return sentences


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


def _map_leaves(fn, x):
    """fn over the leaves of nested dicts / lists / tuples (lists come back as lists, as the reference returns them)."""
    if isinstance(x, dict):
        return {k: _map_leaves(fn, v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_map_leaves(fn, v) for v in x]
    return fn(x)


def var_wrapper(x, cuda=True, volatile=False):
    """misc/utils.py:72-87: numpy arrays / tensors inside nested containers -> tensors on the device (`cuda`) or the
    host; everything else is passed through.  torch.autograd.Variable and `volatile` no longer exist: a volatile
    request yields detached tensors (callers also run under torch.no_grad())."""
    def leaf(v):
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if not torch.is_tensor(v):
            return v
        v = v.cuda() if cuda else v.cpu()
        return v.detach() if volatile else v
    return _map_leaves(leaf, x)


def load_state_dict(model, state_dict):
    """Tolerant checkpoint loader with the behaviour of misc/utils.py:89-107: keys on one side only are reported and
    skipped; a tensor whose shape differs is copied element for element over the common leading part of the two
    flattened tensors (a vocabulary that grew keeps its trained rows)."""
    own = model.state_dict()
    for k in sorted(own.keys() - state_dict.keys()):
        print(f'key {k} in model.state_dict() not in loaded state_dict')
    for k in sorted(state_dict.keys() - own.keys()):
        print(f'key {k} in loaded state_dict not in model.state_dict()')
    with torch.no_grad():
        for k in own.keys() & state_dict.keys():
            src, dst = state_dict[k], own[k]
            if src.shape != dst.shape:
                print(f'key {k} size not match in model.state_dict() and loaded state_dict. '
                      f'Try to flatten and copy the values in common parts')
            n = min(src.numel(), dst.numel())
            dst.reshape(-1)[:n].copy_(src.reshape(-1)[:n])
    model.load_state_dict(own)
