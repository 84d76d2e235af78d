#!/usr/bin/env python3
"""The other BASELINE configs (parity-test cases, not bench.py lines) timed the same way as the headline step:
  C2  att2in2 MLE, B = 64                         (caption_loss_weight 1, everything else 0)
  C4  joint REINFORCE (gt baseline) + self-critical CIDEr-D, B = 256, speaker turn and listener turn
  C3  the headline joint gumbel step, B = 128 (for reference)
usage: config_bench.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build of the library: dispatch switches, stamps)
import torch
from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic
from cooperativeimagecaptioning_amd.misc import rewards


def run(name, steps, turns, **kw):
    opt = synthetic.default_opt(**kw)
    torch.manual_seed(0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt).cuda().train()
    od = optim.load_optimizer(model, opt)
    optim.fuse_zero_grad(od)                     # as train.py and bench.py: the clamp+Adam kernels clear the gradients
    batch = synthetic.make_batch(opt, seed=1234, device='cuda')

    def step(turn):
        o = od[turn] if turn in od else od[list(od)[0]]
        optim.zeroing_optimizer(opt, od, o)
        if opt.is_alternating:
            loss = model(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], None,
                         is_alternating=True, alternating_turn=turn)
        else:
            loss = model(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], None)
        loss.backward()
        optim.update_optimizer(od, o, opt)
        return loss
    for turn in turns:
        for _ in range(3):
            loss = step(turn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(turn)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f'{name:40s} turn {str(turn):9s} B {opt.batch_size:4d}: {dt * 1e3:7.2f} ms/step = {opt.batch_size / dt:8.0f} images/s'
              f'   loss {float(loss.detach()):.4f}', flush=True)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2] if len(sys.argv) > 2 else None      # 'c2bf16': that configuration alone (for a kernel trace of it)
    if os.environ.get('CIC_GEMM_FLAGS') is not None:       # A/B measurement of the GEMM dispatch switches
        from cooperativeimagecaptioning_amd import engine
        engine.lib.cic_debug_gemm_tail_split(int(os.environ['CIC_GEMM_FLAGS'], 0))
    if only == 'c2bf16':
        run('C2 att2in2 MLE (--compute_dtype bf16)', steps, [None], batch_size=64, is_alternating=0, phase=2, caption_loss_weight=1.0,
            retrieval_reward_weight=0.0, cider_optimization=0, alternating_turn=None, compute_dtype='bf16')
        return
    if only == 'c4':
        for _ in range(2):
            run('C4 joint reinforce(gt) + CIDEr-D', steps, ['speaker', 'listener'], batch_size=256, retrieval_reward='reinforce',
                reinforce_baseline_type='gt', vse_loss_weight=1.0)
        return
    run('C3 joint gumbel + CIDEr-D', steps, ['speaker'], batch_size=128)
    run('C3 joint gumbel (--compute_dtype bf16)', steps, ['speaker'], batch_size=128, compute_dtype='bf16')
    run('C2 att2in2 MLE (f32)', steps, [None], batch_size=64, is_alternating=0, phase=2, caption_loss_weight=1.0,
        retrieval_reward_weight=0.0, cider_optimization=0, alternating_turn=None)
    run('C2 att2in2 MLE (--compute_dtype bf16)', steps, [None], batch_size=64, is_alternating=0, phase=2, caption_loss_weight=1.0,
        retrieval_reward_weight=0.0, cider_optimization=0, alternating_turn=None, compute_dtype='bf16')
    run('C4 joint reinforce(gt) + CIDEr-D', steps, ['speaker', 'listener'], batch_size=256, retrieval_reward='reinforce',
        reinforce_baseline_type='gt', vse_loss_weight=1.0)


if __name__ == '__main__':
    main()
