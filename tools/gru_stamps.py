#!/usr/bin/env python3
"""Where a step of the one-launch GRU BPTT loop (gru_seq_bwd_kernel) spends its time: s_memrealtime stamps of lane 0 of every
workgroup (development build, cic_debug_set_gru_stamps), medians over workgroups and steps."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import numpy as np
import torch
from cooperativeimagecaptioning_amd import engine, _lib

lib = _lib.lib
lib.cic_debug_set_gru_stamps.argtypes = [C.c_void_p]
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(3)
B, F, E, J, V, T = 128, 2048, 512, 1024, 9487, 16
Lp = T + 1


def U(*shape, r=0.05):
    return ((torch.rand(*shape, generator=g) * 2 - 1) * r).to(dev)


W = {'img_enc.fc.weight': U(J, F, r=0.03), 'img_enc.fc.bias': U(J), 'txt_enc.embed.weight': U(V + 2, E, r=0.1),
     'txt_enc.rnn.weight_ih_l0': U(3 * J, E), 'txt_enc.rnn.weight_hh_l0': U(3 * J, J, r=0.06),
     'txt_enc.rnn.bias_ih_l0': U(3 * J), 'txt_enc.rnn.bias_hh_l0': U(3 * J)}
params = engine.listener_params(W)
fc = torch.randn(B, F, generator=g).abs().to(dev)
seq = torch.randint(1, V + 1, (B, T), generator=g, dtype=torch.int32).to(dev)       # full-length captions (the bench's case)
stv = torch.ones(B, T, device=dev)
Lt = torch.tensor([T], dtype=torch.int32, device=dev)
dims = engine.listener_dims(B, F, E, J, V, T, Lp)
grads = {k: torch.zeros_like(v) for k, v in W.items()}
gs = torch.ones(1, device=dev)
f = engine.listener_fwd(dims, params, fc, seq=seq, stv=stv, L=Lt)
for _ in range(3):
    engine.listener_bwd(dims, params, f, g_scalar=gs, grads=grads)
nwg = (B // 32) * (J // 16)
buf = torch.zeros(nwg * Lp * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
lib.cic_debug_set_gru_stamps(buf.data_ptr())
engine.listener_bwd(dims, params, f, g_scalar=gs, grads=grads)
torch.cuda.synchronize()
lib.cic_debug_set_gru_stamps(None)
s = buf.cpu().numpy().reshape(nwg, Lp, 8).astype(np.float64) * 0.01     # us
t0 = s[:, Lp - 1, 0].min()
print('kernel start spread (us): max %.2f' % (s[:, Lp - 1, 0].max() - t0))
names = ['gate derivative + stores', 'drain + barrier', 'wait for the strip (poll)', 'A loads + MFMA chain', 'sum barrier', 'sum + loop']
for t in (Lp - 1, Lp - 2, 8, 2, 1):
    row = s[:, t, :]
    d = [row[:, 1] - row[:, 0], row[:, 2] - row[:, 1], row[:, 3] - row[:, 2], row[:, 4] - row[:, 3], row[:, 5] - row[:, 4]]
    nxt = s[:, t - 1, 0] - row[:, 5]
    print('step %2d: ' % t + '  '.join('%s %.2f (p90 %.2f)' % (n, np.median(x), np.percentile(x, 90)) for n, x in zip(names, d + [nxt])))
per = (s[:, 1, 0] - s[:, Lp - 1, 0]) / (Lp - 2)
print('per step (us): median %.2f' % np.median(per), ' whole loop %.1f us' % (s[:, 0, 1].max() - t0))
