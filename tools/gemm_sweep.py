#!/usr/bin/env python3
"""Large-GEMM shapes of one joint step (B=128): time each with the tile shape forced to 128x128 / 64x64 and the
K-sliced tail on / off (cic_debug_gemm_tail_split), HIP events over back-to-back launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch
from cooperativeimagecaptioning_amd import ops, _lib


def timeit(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [  # name, M, N, K, a_kc, b_kc, accumulate (every shape below the three forward ones is a gradient product)
    ('att_embed fwd', 4608, 512, 2048, 1, 1, 0), ('ctx2att fwd', 4608, 512, 512, 1, 1, 0), ('lst gi', 2176, 3072, 512, 1, 1, 0),
    ('d_out', 2048, 512, 9488, 1, 0, 0), ('dW logit', 9488, 512, 2048, 0, 0, 1), ('dW i2h/h2h', 2560, 512, 2048, 0, 0, 1),
    ('dW a2c', 1024, 512, 2048, 0, 0, 1), ('dW h2att', 512, 512, 2048, 0, 0, 1), ('dx', 2048, 512, 2560, 1, 0, 0),
    ('d_att ctx2att', 4608, 512, 512, 1, 0, 1), ('dW ctx2att', 512, 512, 4608, 0, 0, 1), ('dW att_embed', 512, 2048, 4608, 0, 0, 1),
    ('lst d_onehot', 2176, 9488, 512, 1, 1, 0), ('lst dx_emb', 2176, 512, 3072, 1, 0, 0), ('lst dW hh', 3072, 1024, 2176, 0, 0, 1),
    ('lst dW ih', 3072, 512, 2176, 0, 0, 1), ('lst img fc', 128, 1024, 2048, 1, 1, 0), ('lst dW img', 1024, 2048, 128, 0, 0, 1)]


def main():
    dev = 'cuda'
    tot = {}
    for name, M, N, K, akc, bkc, acc in SHAPES:
        A = torch.randn((M, K) if akc else (K, M), device=dev)
        B = torch.randn((N, K) if bkc else (K, N), device=dev)
        C = torch.zeros(M, N, device=dev)
        ref = (A if akc else A.t()).double() @ (B.t() if bkc else B).double()
        row = []
        free = not name.endswith('fwd') and name != 'lst gi' and name != 'lst img fc'
        for tile in (1, 2, 3, 4):
            for split in (0, 1):
                _lib.lib.cic_debug_gemm_tail_split(split | (tile << 8))
                C.zero_()
                ops.gemm(A, B, C, bool(akc), bool(bkc), accumulate=False, sum_order_free=free)
                err = float((C.double() - ref).abs().max() / ref.abs().max())
                us = timeit(lambda: ops.gemm(A, B, C, bool(akc), bool(bkc), accumulate=bool(acc), sum_order_free=free))
                row.append((us, err))
                tot[(tile, split)] = tot.get((tile, split), 0.0) + us
        _lib.lib.cic_debug_gemm_tail_split(1)
        us_auto = timeit(lambda: ops.gemm(A, B, C, bool(akc), bool(bkc), accumulate=bool(acc), sum_order_free=free))
        tot['auto'] = tot.get('auto', 0.0) + us_auto
        fl = 2.0 * M * N * K
        print(f'{name:16s} M{M:5d} N{N:5d} K{K:5d}  128:{row[0][0]:7.1f} 128+tail:{row[1][0]:7.1f}  64:{row[2][0]:7.1f} 64+tail:{row[3][0]:7.1f}'
              f'  128/8w(2x4):{row[4][0]:7.1f} +tail:{row[5][0]:7.1f}  128/8w(4x2):{row[6][0]:7.1f} +tail:{row[7][0]:7.1f}'
              f'  auto:{us_auto:7.1f} us = {fl / us_auto / 1e6:6.1f} TF/s  (peak-time {fl / 157e6:6.1f} us)  maxrelerr {max(r[1] for r in row):.1e}')
    print('totals', {str(k): round(v, 1) for k, v in tot.items()})


if __name__ == '__main__':
    main()
