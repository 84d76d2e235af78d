#!/usr/bin/env python3
"""A/B timing of the bench step under a dispatch switch word of the development build (include/cic_dev.h,
cic_debug_gemm_tail_split): usage: switch_bench.py <word> [<word> ...]   e.g. 1  0x8000001 (bit 27: no fused epilogue)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import torch
from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic
from cooperativeimagecaptioning_amd._lib import lib
from cooperativeimagecaptioning_amd.misc import rewards


def main():
    # a word may carry further switches after commas: gru=<0|1|2> (cic_debug_gru_fused), stop=<0|1> (early stop of the decode
    # and BPTT loops), attn=<0|1> (attention + att2ctx + cell of a decode step as one launch), e.g.  1  1,gru=1  1,attn=0
    specs = sys.argv[1:] or ['1']
    opt = synthetic.default_opt(batch_size=128)
    torch.manual_seed(0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt).cuda().train()
    od = optim.load_optimizer(model, opt)
    o = od['speaker']
    b = synthetic.make_batch(opt, seed=1, device='cuda')
    for f in ('cic_debug_gemm_tail_split', 'cic_debug_gru_fused', 'cic_debug_early_stop', 'cic_debug_bptt_early_stop',
              'cic_debug_attn_cell_fused'):
        getattr(lib, f).argtypes = [C.c_int]
    from cooperativeimagecaptioning_amd.optimizer import fuse_zero_grad
    fuse_zero_grad(od)

    def run(n):
        for _ in range(n):
            optim.zeroing_optimizer(opt, od, o)
            loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True,
                         alternating_turn='speaker')
            loss.backward()
            optim.update_optimizer(od, o, opt)
    for rep in range(3):
        for spec in specs:
            parts = spec.split(',')
            kv = dict(p.split('=') for p in parts[1:])
            lib.cic_debug_gemm_tail_split(int(parts[0], 0))
            lib.cic_debug_gru_fused(int(kv.get('gru', 2)))
            lib.cic_debug_early_stop(int(kv.get('stop', 1)))
            lib.cic_debug_bptt_early_stop(int(kv.get('stop', 1)))
            lib.cic_debug_attn_cell_fused(int(kv.get('attn', 1)))
            run(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(40)
            torch.cuda.synchronize()
            print(f'switches {spec}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step', flush=True)


if __name__ == '__main__':
    main()
