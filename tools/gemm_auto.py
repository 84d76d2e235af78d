#!/usr/bin/env python3
"""The large GEMM shapes of one joint step (tools/gemm_sweep.py's list) as cic_gemm_f32 dispatches them, with the
bf16-part kernel (f32 accuracy on the bf16 matrix cores, csrc/gemm.hip gemm_bfx_kernel) and with the f32-input MFMA
kernel only (development-build switch, bit 28)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import torch
from cooperativeimagecaptioning_amd import ops, _lib
from gemm_sweep import SHAPES, timeit


def main():
    dev = 'cuda'
    tot = {'bf16x3': 0.0, 'f32_mfma': 0.0}
    for name, M, N, K, akc, bkc, acc in SHAPES:
        A = torch.randn((M, K) if akc else (K, M), device=dev)
        B = torch.randn((N, K) if bkc else (K, N), device=dev)
        C = torch.zeros(M, N, device=dev)
        free = not name.endswith('fwd') and name != 'lst gi' and name != 'lst img fc'
        row = {}
        for tag, flags in (('bf16x3', 0x1), ('f32_mfma', 0x10000001)):
            _lib.lib.cic_debug_gemm_tail_split(flags)
            row[tag] = timeit(lambda: ops.gemm(A, B, C, bool(akc), bool(bkc), accumulate=bool(acc), sum_order_free=free))
            tot[tag] += row[tag]
        fl = 2.0 * M * N * K
        print(f'{name:16s} M{M:5d} N{N:5d} K{K:5d}  bf16x3 {row["bf16x3"]:7.1f} us = {fl / row["bf16x3"] / 1e6:6.1f} TF/s(f32-equiv)   '
              f'f32 MFMA {row["f32_mfma"]:7.1f} us = {fl / row["f32_mfma"] / 1e6:6.1f} TF/s', flush=True)
    _lib.lib.cic_debug_gemm_tail_split(1)
    print('totals', {k: round(v, 1) for k, v in tot.items()})


if __name__ == '__main__':
    main()
