#!/usr/bin/env python3
"""Per-timestep products of the joint step (M = batch rows of one decode or of a decode pair): average launch
duration from cic_gemm_f32_timed (back-to-back launches, HIP events in C++), strip walkers on / off."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch
from cooperativeimagecaptioning_amd import _lib
from cooperativeimagecaptioning_amd._lib import GemmArgs, lib

lib.cic_gemm_f32_timed.argtypes = [C.POINTER(GemmArgs), C.c_int, C.POINTER(C.c_double), C.c_void_p]

SHAPES = [  # name, M, N, K, K2, b_kc, accumulate
    ('h2att', 256, 512, 512, 0, 1, 0), ('i2h+h2h', 256, 2560, 512, 512, 1, 0), ('a2c', 256, 1024, 512, 0, 1, 1),
    ('logit', 256, 9488, 512, 0, 1, 0), ('logit M128', 128, 9488, 512, 0, 1, 0), ('logit M512', 512, 9488, 512, 0, 1, 0),
    ('logit M384', 384, 9488, 512, 0, 1, 0), ('logit M64', 64, 9488, 512, 0, 1, 0), ('i2h+h2h M128', 128, 2560, 512, 512, 1, 0),
    ('gru hh', 128, 3072, 1024, 0, 1, 0), ('dres', 128, 512, 1024, 0, 0, 0), ('dh', 128, 512, 2560, 512, 0, 0),
    ('gru dh', 128, 1024, 3072, 0, 0, 1), ('lst img fc', 128, 1024, 2048, 0, 1, 0),
    # the BPTT dX products as the engines launch them (gradient products: K may be split over workgroups), and the
    # same shapes with a K-contiguous (pre-transposed) weight
    ('dres free', 128, 512, 1024, 0, 0, 2), ('dres free kc', 128, 512, 1024, 0, 1, 2),
    ('dh free', 128, 512, 2560, 512, 0, 2), ('dh free kc', 128, 512, 2560, 512, 1, 2),
    ('gru dh free', 128, 1024, 3072, 0, 0, 3), ('gru dh free kc', 128, 1024, 3072, 0, 1, 3)]


def run(name, M, N, K, K2, bkc, acc, flag):
    dev = 'cuda'
    A = torch.randn(M, K, device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cm = torch.zeros(M, N, device=dev)
    g = GemmArgs()
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kc = A.data_ptr(), K, 1
    g.B, g.ldb, g.b_kc = B.data_ptr(), (K if bkc else N), bkc
    ref = A.double() @ (B.t() if bkc else B).double()
    if K2:
        A2 = torch.randn(M, K2, device=dev)
        B2 = torch.randn((N, K2) if bkc else (K2, N), device=dev)
        g.K2, g.A2, g.lda2, g.B2, g.ldb2 = K2, A2.data_ptr(), K2, B2.data_ptr(), (K2 if bkc else N)
        ref = ref + A2.double() @ (B2.t() if bkc else B2).double()
    g.C, g.ldc, g.accumulate = Cm.data_ptr(), N, 0
    if 128 < M <= 256:        # a decode pair: rows [M/2, M) through the second pointer set (same buffers here)
        g.rows_blk = M // 2
        g.A_b, g.C_b = A.data_ptr() + 4 * (M // 2) * K, Cm.data_ptr() + 4 * (M // 2) * N
        if K2:
            g.A2_b = A2.data_ptr() + 4 * (M // 2) * K2
    lib.cic_debug_gemm_tail_split(flag)
    _lib.check(lib.cic_gemm_f32(C.byref(g), None), 'gemm')
    torch.cuda.synchronize()
    err = float((Cm.double() - ref).abs().max() / ref.abs().max())
    g.accumulate = acc & 1
    if acc & 2:
        g.sum_order_free, g.c_is_zero = 1, 1
    us = C.c_double(0)
    _lib.check(lib.cic_gemm_f32_timed(C.byref(g), 200, C.byref(us), None), 'timed')
    lib.cic_debug_gemm_tail_split(1)
    return us.value, err


def main():
    for sh in SHAPES:
        name, M, N, K, K2 = sh[:5]
        fl = 2.0 * M * N * (K + K2)
        u1, e1 = run(*sh, flag=1)
        u0, e0 = run(*sh, flag=1 | (1 << 16) | (1 << 21) | (1 << 22))
        u2, e2 = run(*sh, flag=1 | (1 << 21) | (1 << 22))
        u3, e3 = run(*sh, flag=1 | (1 << 22))
        u4, e4 = run(*sh, flag=1 | (1 << 24))
        u5, e5 = run(*sh, flag=1 | (1 << 25))
        u6, e6 = run(*sh, flag=1 | (2 << 25))
        print(f'{name:14s} M{M:4d} N{N:5d} K{K + K2:5d}  auto {u1:7.2f} us ({fl / u1 / 1e6:6.1f} TF/s, err {e1:.1e})   '
              f'4-wave ldsb {u5:7.2f} us   16-wave ldsb {u6:7.2f} us (err {e6:.1e})   no-rega2 {u4:7.2f} us   no-ldsb {u3:7.2f} us   walk32 {u2:7.2f} us   no-walk {u0:7.2f} us (err {e0:.1e})   '
              f'mfma floor {fl / 157e6:5.2f} us')


if __name__ == '__main__':
    main()
