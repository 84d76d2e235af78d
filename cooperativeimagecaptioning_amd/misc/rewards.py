"""Mirror of the reference's misc/rewards.py:22-72 — the self-critical CIDEr-D reward — on the
GPU (cic_ciderd_reward).  Only df='corpus' exists here, as in the reference checkout (the
pre-computed document-frequency pickles are not part of the repository)."""
import numpy as np
import torch

from .. import engine

CiderD_scorer = None


def init_scorer(cached_tokens):
    """rewards.py:22-24.  The reference builds CiderD(df=cached_tokens); 'corpus' computes the
    document frequencies from the batch, which is what the HIP kernels do."""
    global CiderD_scorer
    if cached_tokens != 'corpus':
        raise NotImplementedError(f"cached_tokens='{cached_tokens}': only the 'corpus' document frequency "
                                  f"(the opts.py default) is available")
    CiderD_scorer = CiderD_scorer or 'corpus'


def get_self_critical_reward_device(refs, ref_off, sample, greedy, ws=None):
    """Device-resident form used by the joint model: DecodeResults in, device tensors out, no sync."""
    return engine.ciderd_reward(sample.seq, sample.L, greedy.seq, greedy.L, refs, ref_off, ws=ws,
                                vocab_size=sample.dims.V)


def _as_dev_i32(x, T=16):
    x = torch.as_tensor(x)
    B, L = x.shape
    out = torch.zeros(B, T, dtype=torch.int32, device='cuda')
    out[:, :L] = x.to('cuda').int()
    return out, torch.tensor([L], dtype=torch.int32, device='cuda')


def get_self_critical_reward(data, gen_result, greedy_res, return_gen_scores=False):
    """rewards.py:34-72 with the reference's signature: tensors [B,L] / [B,L'] in, numpy out."""
    gen, Lg = _as_dev_i32(gen_result)
    gre, Lr = _as_dev_i32(greedy_res)
    refs, ref_off = engine.pack_refs(data['gts'], 'cuda')
    out = engine.ciderd_reward(gen, Lg, gre, Lr, refs, ref_off)
    B = gen.shape[0]
    scores = out['scores'].cpu().numpy()
    cider_greedy = scores[B:].mean()
    diff = scores[:B] - scores[B:]
    if not return_gen_scores:
        return diff, cider_greedy
    return scores[:B], diff, cider_greedy
