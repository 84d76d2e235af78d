#!/usr/bin/env python3
"""In-kernel phase stamps of the row sampler (log-softmax + Gumbel-max + bookkeeping), launched right behind the logit
product as in the decode loop (diagnostic)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch, numpy as np
from cooperativeimagecaptioning_amd import ops, _lib
lib = _lib.lib
lib.cic_debug_set_attn_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
B_, V1, H = 256, 9488, 512
W = torch.randn(V1, H, device=dev) * 0.05
x = torch.randn(B_, H, device=dev)
U = torch.rand(B_, V1, device=dev)
logits = torch.empty(B_, V1, device=dev)
unf = torch.ones(B_, dtype=torch.int32, device=dev); itn = torch.zeros(B_, dtype=torch.int32, device=dev)
seq = torch.zeros(B_, 16, dtype=torch.int32, device=dev); slp = torch.zeros(B_, 16, device=dev); stv = torch.zeros(B_, 16, device=dev)
anyu = torch.zeros(18, dtype=torch.int32, device=dev)
buf = torch.zeros(B_ * 16 * 8, dtype=torch.int64, device=dev)
GUMBEL_ST = 3
def run():
    ops.gemm(x, W, logits)
    ops.logsoftmax_sample(logits, GUMBEL_ST, 1.0, U=U, step=1, unfinished=unf, it_next=itn, seq=seq, slp=slp, stv=stv, any_unfinished=anyu)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.cic_debug_set_attn_stamps(buf.data_ptr())
run()
torch.cuda.synchronize()
lib.cic_debug_set_attn_stamps(None)
s = buf.cpu().numpy().reshape(B_, 16, 8).astype(np.float64)
t0 = s[:, :, 0].min()
rel = (s - t0) * 0.01
names = ['start -> row + noise loaded, local max', 'block max', 'exp + block sum', 'log-probs stored, gumbel + local argmax', 'block argmax', 'exp(z) + block sum3', 'embedding + bookkeeping']
print('wave start us: median %.2f max %.2f;  end median %.2f max %.2f' % (np.median(rel[:, :, 0]), rel[:, :, 0].max(), np.median(rel[:, :, 7]), rel[:, :, 7].max()))
for i, nm in enumerate(names):
    d = rel[:, :, i + 1] - rel[:, :, i]
    print('  %-44s median %.2f us  p90 %.2f  max %.2f' % (nm, np.median(d), np.percentile(d, 90), d.max()))
