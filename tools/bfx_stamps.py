#!/usr/bin/env python3
"""Where the K loop of the batched bf16-part GEMM (gemm_bfx_kernel) spends its time: cumulative s_memrealtime phase times per
wave (development build, cic_debug_set_bfx_stamps) for a few of the step's shapes."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import numpy as np
import torch
from cooperativeimagecaptioning_amd import ops, _lib
lib = _lib.lib
lib.cic_debug_set_bfx_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
SHAPES = [('d_out dlogits W', 2048, 512, 9488, True, False, True), ('dW hh dgh^T h', 3072, 1024, 2176, False, False, True),
          ('d_onehot dx E^T', 2048, 9488, 512, True, True, False), ('att_embed fwd', 4608, 512, 2048, True, True, False)]
if len(sys.argv) > 1 and sys.argv[1] == '--occupancy':
    # the same product at 2.3 rounds of 2 workgroups per CU, at ~1.2 workgroups per CU, and on 150 / 75 CUs only: does a wave's K tile
    # get faster when fewer workgroups ask for operands at the same time?
    SHAPES = [('d_onehot 1200 tiles', 2048, 9488, 512, True, True, False), ('d_onehot 300 tiles', 512, 9488, 512, True, True, False),
              ('d_onehot 150 tiles', 256, 9488, 512, True, True, False), ('d_onehot 75 tiles', 128, 9488, 512, True, True, False)]
for name, M, N, K, akc, bkc, free in SHAPES:
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cc = torch.zeros(M, N, device=dev)
    for _ in range(3):
        ops.gemm(A, B, Cc, akc, bkc, sum_order_free=free)
    buf = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.cic_debug_set_bfx_stamps(buf.data_ptr())
    ops.gemm(A, B, Cc, akc, bkc, sum_order_free=free)
    torch.cuda.synchronize()
    lib.cic_debug_set_bfx_stamps(None)
    s = buf.cpu().numpy().reshape(4096, 8, 8).astype(np.float64)
    used = s[:, :, 7] > 0
    if not used.any():
        print(f'{name}: M{M} N{N} K{K}: not dispatched to gemm_bfx_kernel')
        continue
    w = s[used]                       # [waves, 8]
    nk = w[:, 5]
    per = w[:, :5] * 10.0 / nk[:, None]      # ns per K tile
    span = (s[:, :, 7][used].max() - s[:, :, 6][used].min()) * 0.01
    life = (w[:, 7] - w[:, 6]) * 0.01
    names = ['wait barrier 1', 'store (load wait + split + ds_write)', 'wait barrier 2', 'issue next loads', 'LDS reads + MFMAs']
    print(f'{name}: M{M} N{N} K{K}: {int(used.sum())} waves, K tiles per wave median {np.median(nk):.0f}, kernel span {span:.1f} us, wave lifetime median {np.median(life):.1f} us')
    print('   per K tile (ns): ' + '  '.join(f'{n} {np.median(per[:, i]):.0f}' for i, n in enumerate(names)) + f'  | sum {np.median(per.sum(1)):.0f}')
