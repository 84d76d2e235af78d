#!/usr/bin/env python3
"""In-kernel phase stamps of the attention kernel (diagnostic)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch, numpy as np
from cooperativeimagecaptioning_amd import ops, _lib
lib = _lib.lib
lib.cic_debug_set_attn_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
K, H = 36, 512
W = torch.randn(9488, 512, device=dev)
for B_ in (128, 256):
    att_h = torch.randn(B_, H, device=dev); p_att = torch.randn(B_, K, H, device=dev); att = torch.randn(B_, K, H, device=dev)
    w = torch.randn(H, device=dev); ba = torch.zeros(1, device=dev)
    res, al, dot = torch.empty(B_, H, device=dev), torch.empty(B_, K, device=dev), torch.empty(B_, K, device=dev)
    x = torch.randn(B_, 512, device=dev); out = torch.empty(B_, 9488, device=dev)
    buf = torch.zeros(B_ * 16 * 5, dtype=torch.int64, device=dev)
    for cold in (True, False):
        for _ in range(3):
            if cold: ops.gemm(x, W, out)
            ops.attn_fwd(att_h, p_att, att, w, ba, None, res, al, dot)
        torch.cuda.synchronize()
        if cold: ops.gemm(x, W, out)
        lib.cic_debug_set_attn_stamps(buf.data_ptr())
        ops.attn_fwd(att_h, p_att, att, w, ba, None, res, al, dot)
        torch.cuda.synchronize()
        lib.cic_debug_set_attn_stamps(None)
        s = buf.cpu().numpy().reshape(B_, 16, 5).astype(np.float64)
        s = s[:, (s[0, :, 0] > 0)]          # waves that exist (8 or 16 per workgroup)
        t0 = s[:, :, 0].min()
        rel = (s - t0) * 10.0
        print(f'B={B_} after_gemm={cold}: span {rel[:, :, 4].max() / 1e3:.2f} us; wave start median {np.median(rel[:, :, 0]) / 1e3:.2f} max {rel[:, :, 0].max() / 1e3:.2f}')
        d = rel[:, :, 1:5] - rel[:, :, 0:4]
        for i, nm in enumerate(['start -> first region group reduced (loads landed)', 'remaining region groups', 'barrier wait', 'softmax + weighted sum + store']):
            print('   %-52s median %.2f us  p90 %.2f  max %.2f' % (nm, np.median(d[:, :, i]) / 1e3, np.percentile(d[:, :, i], 90) / 1e3, d[:, :, i].max() / 1e3))
