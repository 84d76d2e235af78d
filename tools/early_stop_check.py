#!/usr/bin/env python3
"""Device-side early stop of the decode loops (AttModel.py:401-408: the reference breaks once every caption has ended) on
the development build (cic_debug_early_stop 1 / 0): ONE full-width joint step (B = 128, 36 x 2048 regions, vocabulary 9487,
ST-Gumbel + CIDEr-D, dropout 0.5) whose captions all end within a few steps (logit.bias[0] raised), run with the early stop
on and off from the same weights, batch and noise, AFTER a step on a batch whose captions run the full 16 steps (so the
workspaces hold stale activations of a long decode).  Checked: token ids of both decodes, their lengths L, the loss and every
logged term equal; every parameter gradient equal to float-atomic tolerance.  Also times the step both ways.

  python tools/early_stop_check.py [--bias 10] [--iters 30]     -> one JSON line; exit code 0 when everything is equal"""
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
    lib.cic_debug_early_stop.argtypes = [C.c_int]
    lib.cic_debug_bptt_early_stop.argtypes = [C.c_int]
    dev = torch.device('cuda', 0)
    opt = synthetic.default_opt(batch_size=args.batch)
    rewards.init_scorer('corpus')
    torch.manual_seed(0)
    model = models.AlternatingJointModel(opt).to(dev).train()
    cg = model.caption_generator
    with contextlib.redirect_stdout(sys.stderr):
        od = optim.load_optimizer(model, opt)
    agents = od['speaker']
    long_batch = synthetic.make_batch(opt, seed=11, device=dev)
    batch = synthetic.make_batch(opt, seed=12, device=dev)
    bias0 = float(cg.logit.bias.data[0])

    def step(b, seed, update=False):
        cg.noise.manual_seed(seed)
        optim.zeroing_optimizer(opt, od, od['speaker'])
        loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True,
                     alternating_turn='speaker')
        loss.backward()
        torch.cuda.synchronize()
        d = model.last_decodes
        out = dict(loss=float(loss.detach()), terms={k: float(v) for k, v in model.loss().items()},
                   seq_s=d['sample'].seq.clone(), seq_g=d['greedy'].seq.clone(), L_s=int(d['sample'].L), L_g=int(d['greedy'].L),
                   slp_s=d['sample'].slp.clone(), grads={a: o.flat.grad.clone() for a, o in agents.items()})
        return out

    res = {}
    for on in (0, 1):
        lib.cic_debug_early_stop(on)
        lib.cic_debug_bptt_early_stop(on)
        cg.logit.bias.data[0] = bias0
        step(long_batch, 5)                                   # leaves the slabs of a full-length decode in the workspaces
        cg.logit.bias.data[0] = args.bias
        res[on] = step(batch, 7)
    a, b = res[0], res[1]
    L = a['L_s']
    ok = a['L_s'] == b['L_s'] and a['L_g'] == b['L_g'] and torch.equal(a['seq_s'], b['seq_s']) and torch.equal(a['seq_g'], b['seq_g'])
    ok = ok and torch.equal(a['slp_s'][:, :L], b['slp_s'][:, :L])
    ok = ok and a['loss'] == b['loss'] and a['terms'] == b['terms']
    gerr = 0.0
    for ag in a['grads']:
        ga, gb = a['grads'][ag], b['grads'][ag]
        gerr = max(gerr, float((ga - gb).abs().max() / (ga.abs().max() + 1e-30)))
        ok = ok and bool(torch.isfinite(gb).all())
    ok = ok and gerr < 1e-5 and max(a['L_s'], a['L_g']) < opt.seq_length
    times = {}
    for on in (0, 1):
        lib.cic_debug_early_stop(on)
        lib.cic_debug_bptt_early_stop(on)
        for _ in range(3):
            step(batch, 7)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            cg.noise.manual_seed(7)
            optim.zeroing_optimizer(opt, od, od['speaker'])
            loss = model(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], batch['att_masks'],
                         is_alternating=True, alternating_turn='speaker')
            loss.backward()
        e1.record()
        torch.cuda.synchronize()
        times[on] = e0.elapsed_time(e1) / args.iters
    print(json.dumps(dict(early_stop_check='ok' if ok else 'MISMATCH', L_sampled=a['L_s'], L_greedy=a['L_g'], loss=b['loss'],
                          max_grad_rel_diff=gerr, fwd_bwd_ms={'all_steps': times[0], 'early_stop': times[1]})), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
