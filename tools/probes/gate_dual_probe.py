#!/usr/bin/env python3
"""What would the gate product cost as two K = 512 half products in ONE launch of the 64-row LDS-staged walker?  Emulated with the
existing kernel: a [256 x 512] x [512 x N] product whose N makes every one of the 256 workgroups walk the columns it would walk in the
dual launch (x·i2h: 2560 columns, h·[h2h; h2att]: 3072 columns -> N = 5632: 64 rows x 88 columns per workgroup).  Compared with
today's launch (K = 1024 strip walker, N = 3072 incl. the attention query) and with each half product as a launch of its own."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _devlib  # noqa: F401,E402
import torch  # noqa: E402
from cooperativeimagecaptioning_amd import _lib  # noqa: E402
from cooperativeimagecaptioning_amd._lib import GemmArgs, lib  # noqa: E402

lib.cic_gemm_f32_timed.argtypes = [C.POINTER(GemmArgs), C.c_int, C.POINTER(C.c_double), C.c_void_p]


def timed(M, N, K, K2=0, tail=0):
    dev = 'cuda'
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    Cm = torch.zeros(M, N, device=dev)
    g = GemmArgs()
    g.M, g.N, g.K = M, N + tail, K
    g.A, g.lda, g.a_kc = A.data_ptr(), K, 1
    g.B, g.ldb, g.b_kc = B.data_ptr(), K, 1
    keep = [A, B, Cm]
    if K2:
        A2 = torch.randn(M, K2, device=dev)
        B2 = torch.randn(N, K2, device=dev)
        g.K2, g.A2, g.lda2, g.B2, g.ldb2 = K2, A2.data_ptr(), K2, B2.data_ptr(), K2
        keep += [A2, B2]
    if tail:
        Bt = torch.randn(tail, K2, device=dev)
        Ct = torch.zeros(M, tail, device=dev)
        bt = torch.zeros(tail, device=dev)
        g.n_split, g.B2_tail, g.ldb2_tail, g.bias_tail, g.C_tail, g.ldc_tail = N, Bt.data_ptr(), K2, bt.data_ptr(), Ct.data_ptr(), tail
        g.C_tail_b = Ct.data_ptr() + 4 * (M // 2) * tail
        keep += [Bt, Ct, bt]
    g.C, g.ldc = Cm.data_ptr(), N
    g.rows_blk = M // 2
    g.A_b, g.C_b = A.data_ptr() + 4 * (M // 2) * K, Cm.data_ptr() + 4 * (M // 2) * N
    if K2:
        g.A2_b = keep[3].data_ptr() + 4 * (M // 2) * K2
    us = C.c_double(0)
    _lib.check(lib.cic_gemm_f32_timed(C.byref(g), 300, C.byref(us), None), 'timed')
    return us.value


print(f"today: [256 x (512+512)] x [2560 gates + 512 query], strip walker     {timed(256, 2560, 512, 512, 512):6.2f} us")
print(f"x·i2h alone      [256 x 512] x [512 x 2560]                            {timed(256, 2560, 512):6.2f} us")
print(f"h·[h2h; h2att]   [256 x 512] x [512 x 3072]                            {timed(256, 3072, 512):6.2f} us")
print(f"dual launch emulated: [256 x 512] x [512 x 5632] (88 columns per workgroup) {timed(256, 5632, 512):6.2f} us")
print(f"logit product for scale: [256 x 512] x [512 x 9488]                    {timed(256, 9488, 512):6.2f} us")
