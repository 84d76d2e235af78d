#!/usr/bin/env python3
"""Where a step of the speaker's one-launch BPTT loop (spk_bptt_seq_kernel) spends its time: s_memrealtime stamps of lane 0
of every workgroup (development build, cic_debug_set_bptt_stamps) during one full-width joint step; medians over workgroups."""
import contextlib
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import numpy as np
import torch
from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic, _lib
from cooperativeimagecaptioning_amd.misc import rewards

lib = _lib.lib
lib.cic_debug_set_bptt_stamps.argtypes = [C.c_void_p]
dev = torch.device('cuda', 0)
B, T = 128, 16
opt = synthetic.default_opt(batch_size=B)
rewards.init_scorer('corpus')
torch.manual_seed(0)
model = models.AlternatingJointModel(opt).to(dev).train()
with contextlib.redirect_stdout(sys.stderr):
    od = optim.load_optimizer(model, opt)
b = synthetic.make_batch(opt, seed=12, device=dev)


def step():
    optim.zeroing_optimizer(opt, od, od['speaker'])
    loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True, alternating_turn='speaker')
    loss.backward()


for _ in range(3):
    step()
nwg = (B // 16) * 32
buf = torch.zeros(nwg * T * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
lib.cic_debug_set_bptt_stamps(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.cic_debug_set_bptt_stamps(None)
s = buf.cpu().numpy().reshape(nwg, T, 8).astype(np.float64) * 0.01     # us
# workgroup -> column tile: with 8 strips a strip sits on one XCD (blockIdx % 8) and jt = blockIdx // 8
jt = np.arange(nwg) // 8 if (B // 16) == 8 else np.arange(nwg) % 32
even = (jt % 2) == 0
for t in (T - 1, T - 2, 8, 1):
    row = s[:, t, :]
    def md(x):
        return '%.2f (p90 %.2f)' % (np.median(x), np.percentile(x, 90))
    e, o = row[even], row[~even]
    print('step %2d: cell %s | hand-off 1 %s | (a,b) products + d att_res %s' % (t, md(row[:, 1] - row[:, 0]), md(row[:, 2] - row[:, 1]), md(row[:, 3] - row[:, 2])))
    print('         even workgroups: publish + wait 2 %s | attention %s | publish 3 + (i,f,o) product %s | wait 3 %s' % (
        md(e[:, 4] - e[:, 3]), md(e[:, 5] - e[:, 4]), md(e[:, 6] - e[:, 5]), md(s[even, t, 7] - e[:, 6])))
    print('         odd workgroups: publish 2 + (i,f,o) product %s | wait 3 + h2att + sum %s' % (md(o[:, 6] - o[:, 3]), md(o[:, 7] - o[:, 6])))
per = (s[:, 1, 0] - s[:, T - 1, 0]) / (T - 2)
print('per step (us): median %.2f;  loop %.1f us' % (np.median(per), s[:, 0, 3].max() - s[:, T - 1, 0].min()))
