#!/usr/bin/env python3
"""The teacher-forced recurrence in one launch (spk_teacher_seq_kernel) against the three-launches-per-step form it replaces, on
the development build (cic_debug_teacher_seq 1 / 0): the att2in2 MLE step of BASELINE configs[1] (B = 64) and at B = 128, with
and without ragged region masks, from the same weights, batch and dropout masks.  Loss within 2e-6 relative and every
parameter gradient within 1e-5 of its largest element (the two forms sum the gate products in different orders and on
different instructions: bf16-part MFMAs in the walker, the f32-input MFMA here); uneven load; timing of the whole step.

  python tools/teacher_seq_check.py [--iters 30]     -> one JSON line; exit code 0 when everything agrees"""
import argparse
import contextlib
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import torch  # noqa: E402
from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic, _lib  # noqa: E402
from cooperativeimagecaptioning_amd.misc import rewards  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--batches', default='64,128', help='batch sizes, comma separated (more than 128 rows: several row blocks)')
    args = ap.parse_args()
    lib = _lib.lib
    lib.cic_debug_teacher_seq.argtypes = [C.c_int]
    dev = torch.device('cuda', 0)
    rewards.init_scorer('corpus')
    report, ok = {}, True
    for B in [int(x) for x in args.batches.split(',')]:
        opt = synthetic.default_opt(batch_size=B, is_alternating=0, phase=2, caption_loss_weight=1.0, retrieval_reward_weight=0.0,
                                    cider_optimization=0, alternating_turn=None)
        torch.manual_seed(0)
        model = models.AlternatingJointModel(opt).to(dev).train()
        cg = model.caption_generator
        with contextlib.redirect_stdout(sys.stderr):
            od = optim.load_optimizer(model, opt)
        o = od[list(od)[0]]
        agents = o if isinstance(o, dict) else {'speaker': o}
        batch = synthetic.make_batch(opt, seed=12, device=dev)
        g = torch.Generator().manual_seed(4)
        nreg = torch.randint(20, batch['att_feats'].shape[1] + 1, (B,), generator=g)
        ragged = (torch.arange(batch['att_feats'].shape[1]).unsqueeze(0) < nreg.unsqueeze(1)).float().to(dev)

        def step(masks, update=False):
            cg.noise.manual_seed(7)
            optim.zeroing_optimizer(opt, od, o)
            loss = model(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], masks)
            loss.backward()
            out = dict(loss=float(loss.detach()), grads={a: x.flat.grad.clone() for a, x in agents.items()})
            if update:
                optim.update_optimizer(od, o, opt)
            return out

        for name, masks in (('plain', None), ('ragged_regions', ragged)):
            res = {}
            for on in (0, 1):
                lib.cic_debug_teacher_seq(on)
                res[on] = step(masks)
            torch.cuda.synchronize()
            x, y = res[0], res[1]
            lerr = abs(x['loss'] - y['loss']) / max(abs(x['loss']), 1e-30)
            gerr, finite = 0.0, True
            for ag in x['grads']:
                ga, gb = x['grads'][ag], y['grads'][ag]
                gerr = max(gerr, float((ga - gb).abs().max() / (ga.abs().max() + 1e-30)))
                finite = finite and bool(torch.isfinite(gb).all())
            report[f'B{B}_{name}'] = dict(loss=y['loss'], loss_rel_diff=lerr, max_grad_rel_diff=gerr, finite=finite, n_grad_buffers=len(x['grads']))
            ok = ok and lerr < 2e-6 and gerr < 1e-5 and finite and len(x['grads']) > 0
        # uneven load
        lib.cic_debug_teacher_seq(0)
        ref = step(None)
        lib.cic_debug_teacher_seq(1)
        side = torch.cuda.Stream()
        big_a = torch.randn(64 << 20, device=dev)
        big_b = torch.empty_like(big_a)
        bad = 0
        for i in range(15):
            with torch.cuda.stream(side):
                for _ in range(4):
                    big_b.copy_(big_a)
            r = step(None)
            torch.cuda.synchronize()
            for ag in r['grads']:
                if not float((ref['grads'][ag] - r['grads'][ag]).abs().max()) <= 1e-5 * float(ref['grads'][ag].abs().max()):
                    bad += 1
        report[f'B{B}_uneven_load'] = dict(runs=15, mismatches=bad)
        ok = ok and bad == 0
        del big_a, big_b
        times = {}
        for on in (0, 1):
            lib.cic_debug_teacher_seq(on)
            for _ in range(3):
                step(None, update=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(args.iters):
                step(None, update=True)
            e1.record()
            torch.cuda.synchronize()
            times[on] = e0.elapsed_time(e1) / args.iters
        report[f'B{B}_step_ms'] = {'three_launches_per_step': times[0], 'one_launch': times[1]}
    report['teacher_seq_check'] = 'ok' if ok else 'MISMATCH'
    print(json.dumps(report), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
