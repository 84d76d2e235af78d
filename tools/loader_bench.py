#!/usr/bin/env python3
"""Input pipeline: the joint step fed from HOST batches (the loader contract of dataloader.py:171-245:
numpy arrays, 37.7 MB per B = 128 batch) with the synchronous per-iteration copy of the reference's trainer
(train.py:162-178) vs prefetch.PrefetchLoader (the next batch uploaded on a copy stream under the current step)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic, train as T
from cooperativeimagecaptioning_amd.misc import rewards
from cooperativeimagecaptioning_amd.prefetch import PrefetchLoader


class FixedHostLoader:
    """Four pre-generated host batches served round robin (the cost under test is the hand-over, not numpy's RNG)."""

    def __init__(self, opt, pin=False):
        gen = synthetic.SyntheticLoader(opt, seed=1, pin=pin)
        self.batches = [gen.get_batch('train') for _ in range(4)]
        self.n = 0
        self.vocab_size, self.seq_length = opt.vocab_size, opt.seq_length

    def get_batch(self, split):
        self.n += 1
        return self.batches[self.n % 4]


def main():
    opt = synthetic.default_opt(batch_size=128)
    torch.manual_seed(0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt).cuda().train()
    od = optim.load_optimizer(model, opt)
    dev = torch.device('cuda', 0)

    def run(loader, n=30, log=None):
        for i in range(n + 5):
            if i == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            data = loader.get_batch('train')
            fc, att, am, labels, masks = T.load_data(data, opt, dev)
            optim.zeroing_optimizer(opt, od, od['speaker'])
            loss = model(fc, labels, masks, data, att, am, is_alternating=True, alternating_turn='speaker')
            loss.backward()
            optim.update_optimizer(od, od['speaker'], opt)
            if hasattr(loader, 'prefetch'):
                loader.prefetch()
            if log is None:
                float(loss.detach())             # the reference trainer's per-iteration host sync (train.py:533-535)
            else:                                # this repo's train.py: the log line follows asynchronously (LossLog)
                log.push(dict(iteration=i, epoch=0, turn='speaker', host_s=0.0, to_history=False), loss, model.loss())
                log.pop(lambda meta, value, terms: None)
        if log is not None:
            log.pop(lambda meta, value, terms: None, block_to=0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    class Resident:
        """the bench.py setting: one batch already in HBM"""
        def __init__(self):
            self.b = synthetic.make_batch(opt, seed=1, device=dev)
            self.b['bounds'] = dict(wrapped=False)
        def get_batch(self, split):
            return self.b
    t_res = run(Resident())
    optim.fuse_zero_grad(od)                     # from here on as train.py: gradients cleared inside the clamp+Adam kernels
    t_trainer = run(Resident(), log=T.LossLog(dev))
    t_trainer_pf = None
    pf0 = PrefetchLoader(FixedHostLoader(opt, pin=True), dev)
    t_trainer_pf = run(pf0, log=T.LossLog(dev))
    pf0.close()
    optim.fuse_zero_grad(od, on=False)
    t_sync = run(FixedHostLoader(opt))
    pf = PrefetchLoader(FixedHostLoader(opt), dev)
    t_pf = run(pf)
    staged = pf.pageable_bytes
    pf.close()
    pf2 = PrefetchLoader(FixedHostLoader(opt, pin=True), dev)
    t_pin = run(pf2)
    staged2 = pf2.pageable_bytes
    pf2.close()
    print(f'batch resident in HBM, host sync per iteration:         {t_res * 1e3:.2f} ms/iteration = {128 / t_res:.0f} images/s')
    print(f'  ... as train.py runs the loop (asynchronous log line, gradients cleared in clamp+Adam): {t_trainer * 1e3:.2f} ms/iteration'
          f' resident, {t_trainer_pf * 1e3:.2f} ms/iteration from pinned host batches through PrefetchLoader')
    print(f'pageable host batches, synchronous copy per iteration: {t_sync * 1e3:.2f} ms/iteration = {128 / t_sync:.0f} images/s')
    print(f'pageable host batches through PrefetchLoader:          {t_pf * 1e3:.2f} ms/iteration = {128 / t_pf:.0f} images/s'
          f'   ({staged / 35 / 1e6:.1f} MB from pageable memory per batch)')
    print(f'pinned host batches through PrefetchLoader (DMA only): {t_pin * 1e3:.2f} ms/iteration = {128 / t_pin:.0f} images/s'
          f'   ({staged2 / 35 / 1e6:.3f} MB from pageable memory per batch: the packed reference captions)')


if __name__ == '__main__':
    main()
