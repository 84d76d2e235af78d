#!/usr/bin/env python3
"""Phase stamps of the bf16-part logit walker (gemm_ldsb2bf_walk_kernel, development build): where a launch's time goes.
Per workgroup and wave: start | A fragments loaded and split | first tile staged (barrier) | per tile: k loop done, barrier passed | end.
The product [256 x 512] x [512 x 9488] with the sampler epilogue off (a plain product; the epilogue's slots ride in the k loop) and -
second table - the same launch with pre-cut weights (B_parts), both as a decode pair (two row blocks of 128).
  python tools/ldsb_bf_stamps.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cooperativeimagecaptioning_amd import _lib, ops  # noqa: E402
from cooperativeimagecaptioning_amd._lib import GemmArgs, lib  # noqa: E402

lib.cic_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
M, N, K = 256, 9488, 512
A = torch.randn(M, K, device=dev)
B = torch.randn(N, K, device=dev)
bias = torch.randn(N, device=dev)
Cm = torch.zeros(M, N, device=dev)
parts = torch.empty(3 * N * K, dtype=torch.int16, device=dev)
ops.split_bf16x3_(B, parts)
pollute = torch.randn(48 << 20, device=dev)          # 192 MB: what passes through the caches between two logit launches of a decode


def run(pre, cold):
    g = GemmArgs()
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kc = A.data_ptr(), K, 1
    g.B, g.ldb, g.b_kc = B.data_ptr(), K, 1
    g.C, g.ldc, g.bias = Cm.data_ptr(), N, bias.data_ptr()
    g.rows_blk = M // 2
    g.A_b, g.C_b = A.data_ptr() + 4 * (M // 2) * K, Cm.data_ptr() + 4 * (M // 2) * N
    if pre:
        g.B_parts = parts.data_ptr()
    buf = torch.zeros(256 * 8 * 64, dtype=torch.int64, device=dev)
    for _ in range(3):
        _lib.check(lib.cic_gemm_f32(C.byref(g), None), 'gemm')
    if cold:
        pollute.add_(1.0)
        A.add_(0.0)                                      # the activations are freshly written in the step
    torch.cuda.synchronize()
    lib.cic_debug_set_stamps(buf.data_ptr())
    _lib.check(lib.cic_gemm_f32(C.byref(g), None), 'gemm')
    torch.cuda.synchronize()
    lib.cic_debug_set_stamps(None)
    raw = buf.cpu().numpy().reshape(256, 8, 64).astype(np.float64)
    t0 = raw[:, :, 0][raw[:, :, 0] > 0].min()
    r = np.where(raw > 0, (raw - t0) * 0.01, np.nan)
    end = r[:, :, 63]
    print(f'pre-cut weights {pre}, caches {"cold" if cold else "warm"}: launch span (first wave start -> last wave end) {np.nanmax(end):.2f} us')
    print(f'  wave start          median {np.nanmedian(r[:, :, 0]):5.2f}  max {np.nanmax(r[:, :, 0]):5.2f}')
    print(f'  A fragments ready   median {np.nanmedian(r[:, :, 1] - r[:, :, 0]):5.2f} after start (p90 {np.nanpercentile(r[:, :, 1] - r[:, :, 0], 90):5.2f})')
    print(f'  first tile staged   median {np.nanmedian(r[:, :, 2] - r[:, :, 1]):5.2f} later')
    ntile = ((~np.isnan(r[:, 0, :63])).sum(1) - 3) // 2
    print(f'  tiles per workgroup {ntile.min()}..{ntile.max()}')
    kl, ba = [], []
    for j in range(int(ntile.max())):
        a, b_, c = r[:, :, 2 + 2 * j], r[:, :, 3 + 2 * j], r[:, :, 4 + 2 * j]
        kl.append(b_ - a)
        ba.append(c - b_)
    kl, ba = np.stack(kl), np.stack(ba)
    print(f'  per tile: k loop    median {np.nanmedian(kl):5.2f} (p90 {np.nanpercentile(kl[~np.isnan(kl)], 90):5.2f}); first tile {np.nanmedian(kl[0]):5.2f}, last {np.nanmedian(kl[-2]):5.2f}')
    print(f'            barrier   median {np.nanmedian(ba):5.2f} (p90 {np.nanpercentile(ba[~np.isnan(ba)], 90):5.2f})')
    last = np.nanmax(np.where(np.isnan(r[:, :, :63]), -1, r[:, :, :63]), axis=2)
    print(f'  tail (last barrier -> end) median {np.nanmedian(end - last):5.2f}')


for pre in (True, False):
    for cold in (False, True):
        run(pre, cold)
