#!/usr/bin/env python3
"""The one-launch GRU pass (gru_seq_kernel: W_hh stationary in registers, per-strip hand-offs inside the launch) against the
one-launch-per-step form it replaces, on the development build (cic_debug_gru_fused 2 / 1): listener forward + backward at the
flagship widths (J = 1024, E = 512, V = 9487), ground-truth and generated captions.  The two forms run the same MFMA chains
in the same order, so EVERY output must be equal bit for bit: loss rows, embeddings, and - through the backward pass, which
reads the saved h / gh slabs of every step - all parameter gradients computed from them (those only to float-atomic
tolerance).  Also runs the hand-off under UNEVEN load (a second stream saturating the chip with a streaming kernel while the
pass runs, L1-warm consumers) and times both forms.

  python tools/gru_seq_check.py [--iters 200]        -> one JSON line; exit code 0 when everything is equal"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: dispatch switches)
import torch  # noqa: E402
from cooperativeimagecaptioning_amd import engine, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--batch', type=int, default=128)
    args = ap.parse_args()
    lib = _lib.lib
    lib.cic_debug_gru_fused.argtypes = [C.c_int]
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(3)
    B, F, E, J, V, T = args.batch, 2048, 512, 1024, 9487, 16
    Lp = T + 1

    def U(*shape, r=0.05):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * r).to(dev)
    W = {'img_enc.fc.weight': U(J, F, r=0.03), 'img_enc.fc.bias': U(J), 'txt_enc.embed.weight': U(V + 2, E, r=0.1),
         'txt_enc.rnn.weight_ih_l0': U(3 * J, E), 'txt_enc.rnn.weight_hh_l0': U(3 * J, J, r=0.06),
         'txt_enc.rnn.bias_ih_l0': U(3 * J), 'txt_enc.rnn.bias_hh_l0': U(3 * J)}
    params = engine.listener_params(W)
    fc = torch.randn(B, F, generator=g).abs().to(dev)
    # generated captions of mixed lengths (some rows end early, one runs to the end)
    seq = torch.randint(1, V + 1, (B, T), generator=g, dtype=torch.int32)
    lens = torch.randint(3, T + 1, (B,), generator=g)
    lens[0] = T
    for b in range(B):
        seq[b, lens[b]:] = 0
    seq = seq.to(dev)
    stv = (1.0 + 1e-3 * torch.randn(B, T, generator=g)).to(dev)
    Lt = torch.tensor([T], dtype=torch.int32, device=dev)
    labels = torch.zeros(B, T + 2, dtype=torch.int64)
    labels[:, 1:T + 1] = seq.cpu().long()
    masks = (torch.arange(T + 2).unsqueeze(0) < (lens.unsqueeze(1) + 2)).float()
    labels, masks = labels.to(dev), masks.to(dev)

    def run(form, mode):
        lib.cic_debug_gru_fused(form)
        if mode == 'generated':
            dims = engine.listener_dims(B, F, E, J, V, T, Lp)
            f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=Lt, want_emb=True)
        else:
            dims = engine.listener_dims(B, F, E, J, V, T, T + 2)
            f = engine.listener_fwd(dims, params, fc, labels=labels, masks=masks, want_emb=True)
        grads = {k: torch.zeros_like(v) for k, v in W.items()}
        gs = torch.ones(1, device=dev)
        engine.listener_bwd(dims, params, f, g_scalar=gs, grads=grads)
        torch.cuda.synchronize()
        return {k: f[k].clone() for k in ('loss_rows', 'loss_sum', 'img_emb', 'cap_emb')}, grads

    report = {}
    ok = True
    for mode in ('generated', 'labels'):
        o1, g1 = run(1, mode)
        o2, g2 = run(2, mode)
        eq = {k: bool(torch.equal(o1[k], o2[k])) for k in o1}
        gerr = max(float((g1[k] - g2[k]).abs().max() / (g1[k].abs().max() + 1e-30)) for k in g1)
        finite = all(bool(torch.isfinite(v).all()) for v in o2.values())
        report[mode] = dict(outputs_bit_equal=eq, max_grad_rel_diff=gerr, finite=finite, loss=float(o2['loss_sum']))
        ok = ok and all(eq.values()) and gerr < 1e-5 and finite
    # uneven load: another stream keeps the chip busy with streaming copies while the pass runs, many times over
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device=dev)
    big_b = torch.empty_like(big_a)
    ref, ref_g = run(1, 'generated')
    lib.cic_debug_gru_fused(2)
    dims = engine.listener_dims(B, F, E, J, V, T, Lp)
    bad = 0
    bad_bwd = 0
    gs = torch.ones(1, device=dev)
    khh = 'txt_enc.rnn.weight_hh_l0'
    gscale = float(ref_g[khh].abs().max())
    for i in range(60):
        with torch.cuda.stream(side):
            for _ in range(3):
                big_b.copy_(big_a)
        f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=Lt, want_emb=True)
        grads = {k: torch.zeros_like(v) for k, v in W.items()}
        with torch.cuda.stream(side):
            for _ in range(3):
                big_b.copy_(big_a)
        engine.listener_bwd(dims, params, f, g_scalar=gs, grads=grads)      # the one-launch BPTT loop under the same load
        if i % 10 == 9:
            torch.cuda.synchronize()
        if not torch.equal(f['cap_emb'], ref['cap_emb']):
            bad += 1
        if not float((grads[khh] - ref_g[khh]).abs().max()) < 1e-5 * gscale:
            bad_bwd += 1
    torch.cuda.synchronize()
    report['uneven_load'] = dict(runs=60, mismatches=bad, backward_mismatches=bad_bwd)
    ok = ok and bad == 0 and bad_bwd == 0
    # timing of the forward engine (whole listener forward: the GRU pass is what differs)
    times = {}
    for form in (1, 2):
        lib.cic_debug_gru_fused(form)
        for _ in range(10):
            f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=Lt, want_emb=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(args.iters):
            f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=Lt, want_emb=True, ws=f['ws'])
        b.record()
        torch.cuda.synchronize()
        times[form] = a.elapsed_time(b) * 1e3 / args.iters
    report['listener_fwd_us'] = {'per_step_launches': times[1], 'one_launch': times[2]}
    # ... and of the backward engine (the BPTT loop is what differs)
    grads = {k: torch.zeros_like(v) for k, v in W.items()}
    for form in (1, 2):
        lib.cic_debug_gru_fused(form)
        f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=Lt, want_emb=True)
        for _ in range(5):
            engine.listener_bwd(dims, params, f, g_scalar=gs, grads=grads)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(args.iters):
            engine.listener_bwd(dims, params, f, g_scalar=gs, grads=grads)
        b.record()
        torch.cuda.synchronize()
        times[form] = a.elapsed_time(b) * 1e3 / args.iters
    report['listener_bwd_us'] = {'per_step_launches': times[1], 'one_launch': times[2]}
    report['gru_seq_check'] = 'ok' if ok else 'MISMATCH'
    print(json.dumps(report), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
