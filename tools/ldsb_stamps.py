#!/usr/bin/env python3
"""Phase stamps of the LDS-staged column walker (the logit product): per workgroup and wave, s_memrealtime at start,
after the first tile is staged, then per tile: MFMA chain done / staged + epilogue issued / barrier passed."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch, numpy as np
from cooperativeimagecaptioning_amd import _lib
from cooperativeimagecaptioning_amd._lib import GemmArgs, lib
lib.cic_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
for M in (256, 128):
    N, K = 9488, 512
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    Cm = torch.zeros(M, N, device=dev)
    g = GemmArgs()
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kc = A.data_ptr(), K, 1
    g.B, g.ldb, g.b_kc = B.data_ptr(), K, 1
    g.C, g.ldc, g.bias = Cm.data_ptr(), N, bias.data_ptr()
    if M > 128:
        g.rows_blk = M // 2
        g.A_b, g.C_b = A.data_ptr() + 4 * (M // 2) * K, Cm.data_ptr() + 4 * (M // 2) * N
    nblk = 256
    buf = torch.zeros(nblk * 4 * 64, dtype=torch.int64, device=dev)
    for _ in range(5):
        _lib.check(lib.cic_gemm_f32(C.byref(g), None), 'gemm')
    torch.cuda.synchronize()
    lib.cic_debug_set_stamps(buf.data_ptr())
    _lib.check(lib.cic_gemm_f32(C.byref(g), None), 'gemm')
    torch.cuda.synchronize()
    lib.cic_debug_set_stamps(None)
    ref = A.double() @ B.double().t() + bias.double()
    print('M', M, 'err', float((Cm.double() - ref).abs().max() / ref.abs().max()))
    raw = buf.cpu().numpy().reshape(nblk, 4, 64).astype(np.float64)
    clk = (raw[:, :, 63] - raw[:, :, 61]) / ((raw[:, :, 62] - raw[:, :, 60]) * 10.0)   # shader cycles per ns
    print('  shader clock inside the tile loop: median %.3f GHz (min %.3f max %.3f)' % (np.median(clk), clk.min(), clk.max()))
    s = raw.copy(); s[:, :, 60:] = 0
    t0 = s[:, :, 0][s[:, :, 0] > 0].min()
    r = np.where(s > 0, (s - t0) * 0.01, np.nan)          # us
    print('  start us: median %.2f max %.2f' % (np.nanmedian(r[:, :, 0]), np.nanmax(r[:, :, 0])))
    print('  prologue (A fragments + first tile staged) us: median %.2f max %.2f' % (np.nanmedian(r[:, :, 1] - r[:, :, 0]), np.nanmax(r[:, :, 1] - r[:, :, 0])))
    ntile = (np.sum(~np.isnan(r[:, 0, :]), 1) - 2) // 3
    print('  tiles per workgroup: min %d max %d' % (ntile.min(), ntile.max()))
    for j in range(int(ntile.max())):
        base = 2 + 3 * j
        prev = r[:, :, base - 1]
        mf = r[:, :, base] - prev
        ep = r[:, :, base + 1] - r[:, :, base]
        ba = r[:, :, base + 2] - r[:, :, base + 1]
        print('  tile %2d: MFMA chain %.2f (p90 %.2f)   stage+epilogue %.2f (p90 %.2f)   barrier %.2f (p90 %.2f) us' % (
            j, np.nanmedian(mf), np.nanpercentile(mf, 90), np.nanmedian(ep), np.nanpercentile(ep, 90), np.nanmedian(ba), np.nanpercentile(ba, 90)))
    print('  end us: median %.2f max %.2f' % (np.nanmedian(np.nanmax(r, axis=2)), np.nanmax(r)))
