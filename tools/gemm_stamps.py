#!/usr/bin/env python3
"""In-kernel phase stamps of the register-streaming GEMM (diagnostic build path)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch, numpy as np
from cooperativeimagecaptioning_amd import ops, _lib
lib = _lib.lib
lib.cic_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
for (M, N, K, K2, bkc) in [(128, 512, 512, 0, 1), (256, 512, 512, 0, 1), (256, 1024, 512, 0, 1), (128, 2560, 512, 512, 1), (128, 512, 2560, 512, 0)]:
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if bkc else (K, N), device=dev)
    A2 = torch.randn(M, K2, device=dev) if K2 else None
    B2 = torch.randn((N, K2) if bkc else (K2, N), device=dev) if K2 else None
    Cc = torch.empty(M, N, device=dev)
    nblk = (M // 32) * ((N + 31) // 32)
    buf = torch.zeros(nblk * 16 * 6, dtype=torch.int64, device=dev)
    for _ in range(5):
        ops.gemm(A, B, Cc, True, bool(bkc), A2=A2, B2=B2)
    torch.cuda.synchronize()
    lib.cic_debug_set_stamps(buf.data_ptr())
    ops.gemm(A, B, Cc, True, bool(bkc), A2=A2, B2=B2)
    torch.cuda.synchronize()
    lib.cic_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(nblk, 16, 6).astype(np.float64)
    t0 = s[:, :, 0].min()
    rel = (s[:, :, :5] - t0) * 10.0      # ns (100 MHz ticks)
    print(f'M{M} N{N} K{K}+{K2} bkc{bkc}: blocks {nblk}')
    print('  kernel span (first start -> last end): %.1f us' % (rel[:, :, 4].max() / 1e3))
    print('  block start times us: min %.2f median %.2f max %.2f' % (rel[:, 0, 0].min() / 1e3, np.median(rel[:, 0, 0]) / 1e3, rel[:, 0, 0].max() / 1e3))
    d = rel[:, :, 1:5] - rel[:, :, 0:4]
    for i, nm in enumerate(['start->first chunk MFMAs done', 'rest of MFMA loop + LDS write', 'barrier wait', 'reduce+store']):
        print('  %-32s median %.2f us  p90 %.2f  max %.2f' % (nm, np.median(d[:, :, i]) / 1e3, np.percentile(d[:, :, i], 90) / 1e3, d[:, :, i].max() / 1e3))
    life = rel[:, :, 4].max(1) - rel[:, :, 0].min(1)
    print('  block lifetime us: median %.2f max %.2f' % (np.median(life) / 1e3, life.max() / 1e3))
    tn = (N + 31) // 32
    st = rel[:, 0, 0].reshape(M // 32, tn)
    print('  start time by strip (median over column tiles) us:', np.round(np.median(st, 1) / 1e3, 2).tolist())
    en = rel[:, :, 4].max(1).reshape(M // 32, tn)
    print('  end   time by strip (median over column tiles) us:', np.round(np.median(en, 1) / 1e3, 2).tolist())
    xcc = s[:, 0, 5].astype(int)
    print('  blocks per XCC:', np.bincount(xcc, minlength=8).tolist())
