#!/usr/bin/env python3
"""Evaluation decode (AttModel.sample_beam): images/s of the device beam search at the flagship
widths (36 x 2048 regions, H = 512, vocab 9487, seq_len 16) next to the CPU oracle's restatement of the reference
loop on a small sample of the same images.  usage: beam_bench.py [beam] [batch]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cooperativeimagecaptioning_amd import models, synthetic


def main():
    beam = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    opt = synthetic.default_opt(batch_size=B)
    torch.manual_seed(0)
    cg = models.setup(opt, 'att2in2', 'caption_model')
    sd = {k: v.clone() for k, v in cg.state_dict().items()}
    cg.cuda().eval()
    batch = synthetic.make_batch(opt, seed=1234, device='cuda')
    o = {'beam_size': beam}
    with torch.no_grad():
        for _ in range(3):
            seq, lps = cg.sample(batch['fc_feats'], batch['att_feats'], None, o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            seq, lps = cg.sample(batch['fc_feats'], batch['att_feats'], None, o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f'device beam search: beam {beam}, B {B}: {dt * 1e3:.2f} ms per batch = {B / dt:.0f} images/s')
    # CPU oracle on a sample
    from oracle import speaker as S
    nb = 4
    cfg = dict(vocab_size=opt.vocab_size, seq_length=opt.seq_length, drop_prob_lm=0.0, decoding_constraint=0)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    att = batch['att_feats'][:nb].cpu()
    t0 = time.perf_counter()
    with torch.no_grad():
        s2, l2, _ = S.sample_beam(sd, cfg, att.mean(1), att, None, o)
    dc = time.perf_counter() - t0
    print(f'CPU oracle (reference loop restated, {torch.get_num_threads()} threads): {nb} images in {dc:.2f} s = {nb / dc:.1f} images/s')
    same = bool((seq[:nb].cpu() == s2).all())
    print('tokens equal on the sample:', same, ' max |logp diff|', float((lps[:nb].cpu() - l2).abs().max()))


if __name__ == '__main__':
    main()
