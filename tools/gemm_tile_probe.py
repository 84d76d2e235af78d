#!/usr/bin/env python3
"""128x128-tile GEMM on exactly 1 / 2 / 4 / 8 tiles per CU and K = 256..2048: separates the fixed cost of a tile round
(launch, first loads, 64 KB epilogue) from the steady-state K-loop rate (tail split off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import ctypes as C, torch
from cooperativeimagecaptioning_amd import _lib
from cooperativeimagecaptioning_amd._lib import GemmArgs, lib
lib.cic_gemm_f32_timed.argtypes = [C.POINTER(GemmArgs), C.c_int, C.POINTER(C.c_double), C.c_void_p]
def run(M, N, K, akc=1, bkc=1, tile=1):
    A = torch.randn((M, K) if akc else (K, M), device='cuda'); B = torch.randn((N, K) if bkc else (K, N), device='cuda'); Cm = torch.zeros(M, N, device='cuda')
    g = GemmArgs(); g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kc = A.data_ptr(), (K if akc else M), akc
    g.B, g.ldb, g.b_kc = B.data_ptr(), (K if bkc else N), bkc
    g.C, g.ldc = Cm.data_ptr(), N
    lib.cic_debug_gemm_tail_split(0 | (tile << 8))
    us = C.c_double(0)
    _lib.check(lib.cic_gemm_f32_timed(C.byref(g), 30, C.byref(us), None), 'timed')
    lib.cic_debug_gemm_tail_split(1)
    return us.value
print('128x128 tiles, nt layout: tiles, K -> us, per-tile-round us, TF/s')
for (M, N) in [(2048, 2048), (4096, 2048), (4096, 4096), (8192, 4096)]:
    for K in (256, 512, 1024, 2048):
        us = run(M, N, K)
        tiles = (M // 128) * (N // 128)
        rounds = tiles / 256
        mf = K / 32 * 4096 / 2.4e3   # us of pure MFMA per tile at 2.4 GHz
        print(f'M{M} N{N} K{K}: tiles {tiles} ({rounds:.0f}/CU)  {us:8.1f} us  per round {us / rounds:7.1f} us (pure MFMA {mf:5.1f})  {2.0 * M * N * K / us / 1e6:6.1f} TF/s')
