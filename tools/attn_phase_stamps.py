#!/usr/bin/env python3
"""Duration of the ATTENTION PHASE inside attn_a2c_cell_kernel (attention -> in-launch hand-off -> att2ctx product + cell, one
launch per decode step), from s_memrealtime stamps of the development build - what bench.py's `roofline` prices the phase with,
now that the attention is no longer a launch of its own.

The bench workload (B = 128 joint step, speaker turn, sampled + greedy decode paired: 256 rows per launch) runs `--iters` forward
passes; the stamps a launch leaves are [workgroup][wave][5] = start, first region group reduced, all reduced, past the barrier,
att_res stores issued (100 MHz ticks, one chip-wide clock).  Every forward pass leaves the stamps of its LAST decode step.
Reported: span = latest `stores issued` minus earliest `start` over all 256 workgroups (the chip-wide duration of the phase: what
the algorithmic bytes of a launch are divided by) and the median per-workgroup duration; medians over the forward passes.

  python tools/attn_phase_stamps.py [--iters 12]  ->  one JSON line"""
import argparse
import contextlib
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cooperativeimagecaptioning_amd import models, synthetic, status, _lib  # noqa: E402
from cooperativeimagecaptioning_amd.misc import rewards  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--batch', type=int, default=128)
    args = ap.parse_args()
    lib = _lib.lib
    lib.cic_debug_set_attn_stamps.argtypes = [C.c_void_p]
    dev = torch.device('cuda', 0)
    rewards.init_scorer('corpus')
    opt = synthetic.default_opt(batch_size=args.batch)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = models.AlternatingJointModel(opt).to(dev).train()
    batch = synthetic.make_batch(opt, seed=1, device=dev)
    rows = 2 * args.batch
    buf = torch.zeros(rows * 16 * 5, dtype=torch.int64, device=dev)

    def fwd():
        return model(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], batch['att_masks'],
                     is_alternating=True, alternating_turn='speaker')
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    spans, wg_med, wg_p90 = [], [], []
    for _ in range(args.iters):
        buf.zero_()
        lib.cic_debug_set_attn_stamps(buf.data_ptr())
        fwd()
        torch.cuda.synchronize()
        lib.cic_debug_set_attn_stamps(None)
        s = buf.cpu().numpy().reshape(rows, 16, 5).astype(np.float64)
        if s[:, :, 0].min() <= 0:          # the decode ended before its last step (every caption closed): no launch stamped all rows
            continue
        spans.append((s[:, :, 4].max() - s[:, :, 0].min()) * 0.01)
        per_wg = (s[:, :, 4].max(axis=1) - s[:, :, 0].min(axis=1)) * 0.01
        wg_med.append(float(np.median(per_wg)))
        wg_p90.append(float(np.percentile(per_wg, 90)))
    status.check(dev, 'attn_phase_stamps')
    fused = bool(getattr(model.caption_generator, 'last_pair_fused', False))
    out = dict(kernel='attn_a2c_cell_kernel<5, float>', rows_per_launch=rows, paired=fused, launches_stamped=len(spans),
               phase_span_us=float(np.median(spans)) if spans else None,
               phase_span_us_min=float(np.min(spans)) if spans else None, phase_span_us_max=float(np.max(spans)) if spans else None,
               per_workgroup_median_us=float(np.median(wg_med)) if wg_med else None,
               per_workgroup_p90_us=float(np.median(wg_p90)) if wg_p90 else None,
               clock='s_memrealtime, 100 MHz', what='start of the first wave -> att_res stores issued by the last wave, all workgroups')
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
