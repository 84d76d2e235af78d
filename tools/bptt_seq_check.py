#!/usr/bin/env python3
"""The speaker's one-launch BPTT loop (spk_bptt_seq_kernel: a2c / h2h / h2att columns stationary in the workgroups, three
hand-offs per step inside a 16-row strip) against the four-launches-per-step form it replaces, on the development build
(cic_debug_bptt_seq 1 / 0): ONE full-width joint step (B = 128, 36 x 2048 regions, vocabulary 9487, ST-Gumbel + CIDEr-D,
dropout 0.5) from the same weights, batch and noise - with full-length captions and with captions that all end early
(logit.bias[0] raised: the steps beyond the decode's length) - and with ragged region masks.  The forward pass is the same
code, so tokens and loss must be equal; every parameter gradient must agree to summation-order tolerance (1e-5 of the
gradient's largest element).  Also runs the loop under UNEVEN load (a second stream streaming copies) and times both forms.

  python tools/bptt_seq_check.py [--iters 30]     -> one JSON line; exit code 0 when everything agrees"""
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
    ap.add_argument('--bias', type=float, default=10.0)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--batch', type=int, default=128)
    args = ap.parse_args()
    lib = _lib.lib
    lib.cic_debug_bptt_seq.argtypes = [C.c_int]
    dev = torch.device('cuda', 0)
    opt = synthetic.default_opt(batch_size=args.batch)
    rewards.init_scorer('corpus')
    torch.manual_seed(0)
    model = models.AlternatingJointModel(opt).to(dev).train()
    cg = model.caption_generator
    with contextlib.redirect_stdout(sys.stderr):
        od = optim.load_optimizer(model, opt)
    agents = od['speaker']
    batch = synthetic.make_batch(opt, seed=12, device=dev)
    masked = dict(batch)
    g = torch.Generator().manual_seed(4)
    nreg = torch.randint(20, batch['att_feats'].shape[1] + 1, (args.batch,), generator=g)
    masked['att_masks'] = (torch.arange(batch['att_feats'].shape[1]).unsqueeze(0) < nreg.unsqueeze(1)).float().to(dev)
    bias0 = float(cg.logit.bias.data[0])

    def step(b, seed):
        cg.noise.manual_seed(seed)
        optim.zeroing_optimizer(opt, od, od['speaker'])
        loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True,
                     alternating_turn='speaker')
        loss.backward()
        d = model.last_decodes
        return dict(loss=loss.detach().clone(), seq_s=d['sample'].seq.clone(), L_s=d['sample'].L.clone(),
                    grads={a: o.flat.grad.clone() for a, o in agents.items()})

    report, ok = {}, True
    cases = (('full_length', batch, bias0), ('early_end', batch, args.bias), ('ragged_regions', masked, bias0))
    for name, b, bias in cases:
        res = {}
        for on in (0, 1):
            lib.cic_debug_bptt_seq(on)
            cg.logit.bias.data[0] = bias
            res[on] = step(b, 7)
        torch.cuda.synchronize()
        x, y = res[0], res[1]
        same = bool(torch.equal(x['seq_s'], y['seq_s'])) and float(x['loss']) == float(y['loss'])
        gerr = 0.0
        finite = True
        for ag in x['grads']:
            ga, gb = x['grads'][ag], y['grads'][ag]
            gerr = max(gerr, float((ga - gb).abs().max() / (ga.abs().max() + 1e-30)))
            finite = finite and bool(torch.isfinite(gb).all())
        report[name] = dict(L=int(x['L_s']), forward_equal=same, max_grad_rel_diff=gerr, finite=finite)
        ok = ok and same and finite and gerr < 1e-5
    cg.logit.bias.data[0] = bias0
    # uneven load: a second stream streams copies while the loop runs
    lib.cic_debug_bptt_seq(0)
    ref = step(batch, 7)
    lib.cic_debug_bptt_seq(1)
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device=dev)
    big_b = torch.empty_like(big_a)
    bad = 0
    for i in range(20):
        with torch.cuda.stream(side):
            for _ in range(6):
                big_b.copy_(big_a)
        r = step(batch, 7)
        torch.cuda.synchronize()
        for ag in r['grads']:
            ga, gb = ref['grads'][ag], r['grads'][ag]
            if not float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max()):
                bad += 1
    report['uneven_load'] = dict(runs=20, mismatches=bad)
    ok = ok and bad == 0
    times = {}
    for on in (0, 1):
        lib.cic_debug_bptt_seq(on)
        for _ in range(3):
            step(batch, 7)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.iters):
            step(batch, 7)
        e1.record()
        torch.cuda.synchronize()
        times[on] = e0.elapsed_time(e1) / args.iters
    report['fwd_bwd_ms'] = {'four_launches_per_step': times[0], 'one_launch': times[1]}
    report['bptt_seq_check'] = 'ok' if ok else 'MISMATCH'
    print(json.dumps(report), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
