#!/usr/bin/env python3
"""rocprofv3 probe of the per-step GEMM shapes (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperativeimagecaptioning_amd import ops
dev = 'cuda'
shapes = [(128, 512, 512, 1), (128, 2560, 1024, 1), (128, 1024, 512, 1), (128, 9488, 512, 1), (128, 3072, 1024, 1),
          (128, 512, 3072, 0), (128, 512, 1024, 0), (128, 1024, 3072, 0)]
for M, N, K, bkc in shapes:
    A = torch.randn(M, K, device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(30):
        ops.gemm(A, B, C, True, bool(bkc))
    torch.cuda.synchronize()
