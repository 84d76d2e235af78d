#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (HIP events on torch's current stream)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperativeimagecaptioning_amd import ops


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def main():
    dev = 'cuda'
    print('device', torch.cuda.get_device_name(0))
    shapes = [(128, 512, 512, 1, 1), (128, 2560, 512, 1, 1), (128, 1024, 512, 1, 1), (128, 9488, 512, 1, 1),
              (128, 3072, 1024, 1, 1), (4608, 512, 2048, 1, 1), (4608, 512, 512, 1, 1),
              (2048, 512, 9488, 1, 0), (9488, 512, 2048, 0, 0), (2560, 512, 2048, 0, 0),
              (2048, 9488, 512, 1, 1), (512, 2048, 4608, 0, 0), (128, 512, 2560, 1, 0), (2176, 3072, 512, 1, 1),
              (128, 512, 1024, 1, 0), (128, 512, 3072, 1, 0), (128, 1024, 3072, 1, 0), (4608, 512, 512, 1, 0),
              (512, 512, 4608, 0, 0), (2048, 512, 2560, 1, 0), (512, 512, 2048, 0, 0), (1024, 512, 2048, 0, 0),
              (2176, 512, 3072, 1, 0), (3072, 512, 2176, 0, 0), (3072, 1024, 2176, 0, 0), (2176, 9488, 512, 1, 1),
              (9488, 512, 2176, 0, 0), (128, 128, 1024, 1, 1)]
    for M, N, K, akc, bkc in shapes:
        A = torch.randn((M, K) if akc else (K, M), device=dev)
        B = torch.randn((N, K) if bkc else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        us = timeit(lambda: ops.gemm(A, B, C, bool(akc), bool(bkc)))
        tf = 2.0 * M * N * K / us / 1e6
        ref = (A if akc else A.t()) @ (B.t() if bkc else B)
        us_t = timeit(lambda: torch.mm(A if akc else A.t(), B.t() if bkc else B, out=C))
        print(f'gemm M{M} N{N} K{K} akc{akc} bkc{bkc}: {us:8.1f} us {tf:6.1f} TF/s   (torch.mm {us_t:8.1f} us)')
    for B_ in (128, 256):
        K, H = 36, 512
        att_h = torch.randn(B_, H, device=dev)
        p_att = torch.randn(B_, K, H, device=dev)
        att = torch.randn(B_, K, H, device=dev)
        w = torch.randn(H, device=dev)
        ba = torch.zeros(1, device=dev)
        res, al, dot = torch.empty(B_, H, device=dev), torch.empty(B_, K, device=dev), torch.empty(B_, K, device=dev)
        us = timeit(lambda: ops.attn_fwd(att_h, p_att, att, w, ba, None, res, al, dot), iters=200)
        byts = B_ * 151696
        print(f'attn_fwd B{B_}: {us:6.2f} us  {byts / us / 1e6:6.2f} TB/s algorithmic  frac of 8 TB/s {byts / us / 1e6 / 8:.3f}')


if __name__ == '__main__':
    main()
