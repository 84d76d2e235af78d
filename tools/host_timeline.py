#!/usr/bin/env python3
"""Host-side enqueue time of the three parts of a bench step (no synchronisation inside the loop) against the step's GPU
time: shows whether the host runs ahead of the device or the device waits for launches."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic
from cooperativeimagecaptioning_amd.misc import rewards


def main():
    opt = synthetic.default_opt(batch_size=128)
    torch.manual_seed(0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt).cuda().train()
    od = optim.load_optimizer(model, opt)
    o = od['speaker']
    b = synthetic.make_batch(opt, seed=1, device='cuda')
    n = 40
    acc = [0.0, 0.0, 0.0, 0.0]
    for i in range(n + 5):
        if i == 5:
            torch.cuda.synchronize()
            acc = [0.0, 0.0, 0.0, 0.0]
            w0 = time.perf_counter()
        t0 = time.perf_counter()
        optim.zeroing_optimizer(opt, od, o)
        t1 = time.perf_counter()
        loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True,
                     alternating_turn='speaker')
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        optim.update_optimizer(od, o, opt)
        t4 = time.perf_counter()
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += d
    t_enq = time.perf_counter() - w0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - w0
    print('host enqueue per step: zero_grad %.3f ms, forward %.3f ms, backward %.3f ms, update %.3f ms' %
          tuple(a / n * 1e3 for a in acc))
    print('all launches enqueued after %.3f ms/step; device done after %.3f ms/step' % (t_enq / n * 1e3, t_all / n * 1e3))


if __name__ == '__main__':
    main()
