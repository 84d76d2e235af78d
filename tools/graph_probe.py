#!/usr/bin/env python3
"""Is a launch boundary cheaper from a HIP graph?  The paired decode loop of the bench step (85 launches) timed as eager
launches and as ONE captured graph replayed (timing only: the Philox offsets are baked into the captured arguments)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperativeimagecaptioning_amd import models, synthetic, engine
from cooperativeimagecaptioning_amd.misc import rewards


def main():
    dev = torch.device('cuda', 0)
    opt = synthetic.default_opt(batch_size=128)
    rewards.init_scorer('corpus')
    torch.manual_seed(0)
    model = models.AlternatingJointModel(opt).to(dev).train()
    cg = model.caption_generator
    b = synthetic.make_batch(opt, seed=1, device=dev)
    att_pre = cg.att_embed_pre(b['att_feats'])
    ra = cg._decode_io(b['att_feats'], None, att_pre=att_pre, mode='gumbel', temp=1.0, grad=True)
    rb = cg._decode_io(b['att_feats'], None, att_pre=att_pre, mode='greedy', tag='greedy')
    dims, params, ia = ra[0], ra[1], ra[2]
    ib = rb[2]

    def run():
        engine.speaker_decode_fwd_pair(dims, params, ia, ib)
    for _ in range(5):
        run()
    torch.cuda.synchronize()

    def timeit(f, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    eager = timeit(run, 50)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    graph = timeit(g.replay, 50)
    print(f'paired decode loop: eager {eager * 1e3:.1f} us, graph replay {graph * 1e3:.1f} us')


if __name__ == '__main__':
    main()
