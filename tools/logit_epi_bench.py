#!/usr/bin/env python3
"""Times the logit product of a paired decode step ([2B,512] x [512,9488], the LDS-staged walker) with and without its
fused vocabulary epilogue (row partials for the log-softmax / sampler), per mode of the two row blocks."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import torch
from cooperativeimagecaptioning_amd import _lib
from cooperativeimagecaptioning_amd._lib import GemmArgs, lib


class EpiRows(C.Structure):
    _fields_ = [('mode', C.c_int), ('inv_temp', C.c_float), ('noise', C.c_int), ('U', C.c_void_p), ('ldu', C.c_int),
                ('philox', C.c_int), ('seed', C.c_uint64), ('elem0', C.c_uint64), ('cons_seq', C.c_void_p),
                ('cons_ld', C.c_int), ('cons_col', C.c_int), ('part', C.c_void_p), ('part_rows', C.c_int)]


class Epilogue(C.Structure):
    _fields_ = [('blk', EpiRows * 2)]


lib.cic_gemm_f32_timed.argtypes = [C.POINTER(GemmArgs), C.c_int, C.POINTER(C.c_double), C.c_void_p]
lib.cic_gemm_logit_parts.argtypes = [C.POINTER(GemmArgs)]


def main():
    B, H, V1 = 128, 512, 9488
    dev = 'cuda'
    out_a, out_b = torch.randn(B, H, device=dev), torch.randn(B, H, device=dev)
    W, bias = torch.randn(V1, H, device=dev) * 0.05, torch.randn(V1, device=dev) * 0.1
    la, lb = torch.empty(B, V1, device=dev), torch.empty(B, V1, device=dev)
    U = torch.rand(B, V1, device=dev)
    part_a, part_b = torch.empty(6 * 16384, device=dev), torch.empty(6 * 16384, device=dev)
    g = GemmArgs()
    g.M, g.N, g.K = 2 * B, V1, H
    g.A, g.lda, g.a_kc = out_a.data_ptr(), H, 1
    g.B, g.ldb, g.b_kc = W.data_ptr(), H, 1
    g.C, g.ldc, g.bias = la.data_ptr(), V1, bias.data_ptr()
    g.rows_blk, g.A_b, g.C_b = B, out_b.data_ptr(), lb.data_ptr()
    st = torch.cuda.current_stream().cuda_stream

    def run(tag, epi):
        g.epi = C.addressof(epi) if epi is not None else None
        us = C.c_double(0.0)
        _lib.check(lib.cic_gemm_f32_timed(C.byref(g), 200, C.byref(us), st), tag)
        print(f'{tag:44s} {us.value:7.2f} us   {2.0 * 2 * B * H * V1 / us.value / 1e6:6.1f} TF/s', flush=True)

    def epi(mode_a, noise_a, philox, mode_b=1):
        e = Epilogue()
        for q, (mode, noise, part) in enumerate(((mode_a, noise_a, part_a), (mode_b, 0, part_b))):
            r = e.blk[q]
            r.mode, r.inv_temp, r.noise, r.ldu = mode, 1.0, noise, V1
            if noise and not philox:
                r.U = U.data_ptr()
            r.philox, r.seed, r.elem0 = int(philox), 1234, 4 * (7 << 32)
            r.part, r.part_rows = part.data_ptr(), B
        return e
    if len(sys.argv) > 1:
        lib.cic_debug_gemm_tail_split(int(sys.argv[1], 0))
    print('parts per row:', lib.cic_gemm_logit_parts(C.byref(g)))
    run('no epilogue', None)
    run('epilogue: none / none (m1, s1 only)', epi(0, 0, False, 0))
    run('epilogue: greedy / greedy', epi(1, 0, False))
    run('epilogue: gumbel-ST (no noise) / greedy', epi(3, 0, False))
    run('epilogue: gumbel-ST (U in memory) / greedy', epi(3, 1, False))
    run('epilogue: gumbel-ST (Philox) / greedy', epi(3, 1, True))
    run('epilogue: multinomial (Philox) / greedy', epi(2, 1, True))
    # the weights already cut into their three bf16 parts (cic_split_bf16x3, once per training step)
    from cooperativeimagecaptioning_amd import ops
    ref_a = la.clone()
    parts = torch.empty(3 * W.numel(), dtype=torch.int16, device=dev)
    ops.split_bf16x3_(W, parts)
    g.B_parts = parts.data_ptr()
    run('pre-split W: no epilogue', None)
    run('pre-split W: gumbel-ST (Philox) / greedy', epi(3, 1, True))
    torch.cuda.synchronize()
    print('pre-split logits equal the on-the-fly ones bit for bit:', bool(torch.equal(ref_a, la)))


if __name__ == '__main__':
    main()
