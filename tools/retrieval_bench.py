#!/usr/bin/env python3
"""Retrieval-rank evaluation at the COCO test size (5000 images x 5 captions, 1024-d embeddings): device time of the
similarity GEMM + rank kernels next to the CPU oracle (the reference's per-query np.dot + np.argsort loops) on a
sample of the queries."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cooperativeimagecaptioning_amd import eval_utils as E


def main():
    N, K = 5000, 1024
    rs = np.random.RandomState(0)
    im = rs.randn(N, K).astype(np.float32)
    im /= np.linalg.norm(im, axis=1, keepdims=True)
    cap = 0.15 * np.repeat(im, 5, 0) + rs.randn(5 * N, K).astype(np.float32) / np.sqrt(K)
    cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    ims, caps = torch.from_numpy(im).cuda(), torch.from_numpy(cap).cuda()
    for _ in range(2):
        E.retrieval_ranks(ims, caps, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = E.retrieval_ranks(ims, caps, 5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 2.0 * N * 5 * N * K
    print(f'device: {N} images x {5 * N} captions: {dt * 1e3:.2f} ms (similarity GEMM {fl / 1e9:.0f} GFLOP + i2t + t2i ranks)')
    from oracle import retrieval as R
    images = np.repeat(im, 5, 0)
    nq = 100
    t0 = time.perf_counter()
    R.i2t(images, cap, npts=nq)
    t1 = time.perf_counter()
    R.t2i(images, cap, 5, npts=nq)
    t2 = time.perf_counter()
    est = ((t1 - t0) + (t2 - t1)) * N / nq
    print(f'CPU oracle (reference loops): {nq} of {N} query images in {t2 - t0:.2f} s -> {est:.1f} s for the full evaluation '
          f'({est / dt:.0f}x)')


if __name__ == '__main__':
    main()
