#!/usr/bin/env python3
"""What shader clock does the chip hold (a) idle, (b) right behind the launch-bound decode loop,
(c) behind a dense GEMM stream?  Interprets kernel durations in profiles/."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch
from cooperativeimagecaptioning_amd import ops, _lib, models, synthetic, optimizer as optim
from cooperativeimagecaptioning_amd.misc import rewards

lib = _lib.lib
lib.cic_debug_clock_mhz.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.cic_debug_empty.argtypes = [C.c_int, C.c_int, C.c_void_p]


def probe(tag, spin=200000):
    out = torch.zeros(2, device='cuda')
    lib.cic_debug_clock_mhz(out.data_ptr(), spin, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print(f'{tag}: {float(out[0]):.0f} MHz', flush=True)


torch.cuda.set_device(0)
probe('cold')
time.sleep(1.0)
probe('after 1 s idle')
A = torch.randn(4608, 2048, device='cuda'); B = torch.randn(512, 2048, device='cuda'); Cc = torch.empty(4608, 512, device='cuda')
for _ in range(200):
    ops.gemm(A, B, Cc)
probe('behind 200 big GEMMs')
opt = synthetic.default_opt(batch_size=128)
torch.manual_seed(0)
rewards.init_scorer('corpus')
model = models.AlternatingJointModel(opt).cuda().train()
od = optim.load_optimizer(model, opt)
batch = synthetic.make_batch(opt, device='cuda')
for i in range(6):
    optim.zeroing_optimizer(opt, od, od['speaker'])
    loss = model(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], None, is_alternating=True, alternating_turn='speaker')
    loss.backward()
    optim.update_optimizer(od, od['speaker'], opt)
    probe(f'behind joint step {i}', spin=20000)
# empty-kernel cadence
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream().cuda_stream
for grid, blk in ((128, 1024), (16, 1024), (256, 256)):
    e0.record()
    for _ in range(2000):
        lib.cic_debug_empty(grid, blk, st)
    e1.record()
    torch.cuda.synchronize()
    print(f'empty kernel {grid}x{blk}: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per launch (back-to-back)', flush=True)
