#!/usr/bin/env python3
"""rocprofv3 probe: attention kernel duration vs batch, in a cache-polluting loop like the real step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch
from cooperativeimagecaptioning_amd import ops, _lib
lib = _lib.lib
lib.cic_debug_empty.argtypes = [C.c_int, C.c_int, C.c_void_p]
dev = 'cuda'
K, H = 36, 512
big = torch.randn(64 << 20, device=dev)      # 256 MB streamed between launches: evicts L2 (and most of MALL)
W = torch.randn(9488, 512, device=dev)
for B_ in (64, 128, 256):
    att_h = torch.randn(B_, H, device=dev); p_att = torch.randn(B_, K, H, device=dev); att = torch.randn(B_, K, H, device=dev)
    w = torch.randn(H, device=dev); ba = torch.zeros(1, device=dev)
    res, al, dot = torch.empty(B_, H, device=dev), torch.empty(B_, K, device=dev), torch.empty(B_, K, device=dev)
    x = torch.randn(B_, 512, device=dev); out = torch.empty(B_, 9488, device=dev)
    for it in range(20):
        ops.gemm(x, W, out)                      # like the step: 19 MB of weights between attention launches
        ops.attn_fwd(att_h, p_att, att, w, ba, None, res, al, dot)
    torch.cuda.synchronize()
    for it in range(20):                         # back-to-back, warm
        ops.attn_fwd(att_h, p_att, att, w, ba, None, res, al, dot)
    torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for it in range(50):
    lib.cic_debug_empty(128, 1024, st)
torch.cuda.synchronize()
