#!/usr/bin/env python3
"""Data-parallel equivalence of the joint step:

    N ranks x one micro-batch each, flat gradients summed by the bucketed all-reduce, 1/N folded into clamp+Adam
        ==
    ONE rank that runs the N micro-batches one after the other, accumulates their gradients in the same flat buffers
    and applies clamp+Adam with grad_scale = 1/N

on REAL joint steps (sampled + greedy decode, listener, CIDEr-D reward, both backward engines): same initial weights,
rank r's batch and noise stream in both runs.  Checked: (1) the replicas stay bit-identical, (2) the exchanged gradient
equals the accumulated one to float-atomic tolerance (gradient products sum partial tiles with float atomics, so the
last bits depend on the launch), (3) so do the parameters after every step.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
      tools/dp_equivalence.py --backend gloo --same-device          # rehearsal on ONE GPU (what tests/ runs)
  python -m torch.distributed.run ... --nproc-per-node 8 tools/dp_equivalence.py    # RCCL, one GPU per rank

Exit code 0 and a JSON line on rank 0 when the equality holds.
"""
import argparse
import contextlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--same-device', action='store_true', help='every rank on cuda:0 (gloo rehearsal on a one-GPU box)')
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--small', action='store_true', help='reduced widths (quick rehearsal); default: the flagship widths')
    ap.add_argument('--share-embed', action='store_true', help='share_embed = 1: one table in both agents, its gradient in the listener bucket')
    args = ap.parse_args()
    world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
    local_rank = 0 if args.same_device else int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert not args.same_device or args.backend == 'gloo', '--same-device is a gloo rehearsal (RCCL wants one GPU per rank)'
    if args.same_device:
        os.environ['CIC_SHARED_DEVICE'] = '1'         # several ranks compute on one GPU: no launches that need the whole chip
    torch.cuda.set_device(local_rank)
    dist.init_process_group(args.backend)
    dev = torch.device('cuda', local_rank)

    from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic
    from cooperativeimagecaptioning_amd.misc import rewards
    from cooperativeimagecaptioning_amd.noise import NoiseSource
    kw = dict(batch_size=args.batch)
    if args.small:
        kw.update(vocab_size=199, rnn_size=64, input_encoding_size=64, att_hid_size=64, fc_feat_size=128,
                  att_feat_size=128, vse_embed_size=128)
    if args.share_embed:
        kw.update(share_embed=1)
    rewards.init_scorer('corpus')

    def make():
        opt = synthetic.default_opt(**kw)
        torch.manual_seed(0)                                 # identical initial replicas
        m = models.AlternatingJointModel(opt)
        m.caption_generator.logit.bias.data[0] = 1.0         # captions of different lengths
        m.to(dev).train()
        with contextlib.redirect_stdout(sys.stderr):
            od = optim.load_optimizer(m, opt)
        return opt, m, od

    def fwd_bwd(m, opt, batch):
        loss = m(batch['fc_feats'], batch['labels'], batch['masks'], batch, batch['att_feats'], batch['att_masks'],
                 is_alternating=True, alternating_turn='speaker')
        loss.backward()
        return float(loss.detach())

    K = 9 if args.small else 36
    batches = {(r, s): None for r in range(world) for s in range(args.steps)}

    def batch_of(opt, r, s):
        if batches[(r, s)] is None:
            batches[(r, s)] = synthetic.make_batch(opt, K=K, seed=1234 + 17 * s + r, device=dev)
        return batches[(r, s)]

    # ---- data-parallel run: this rank's micro-batch, bucketed exchange (listener + logit bucket from inside backward)
    opt, model, od = make()
    model.caption_generator.noise = NoiseSource(1000 + rank)
    optim.overlap_gradient_exchange(model, od)
    agents = od['speaker']
    dp_grads, dp_params = [], []
    for s in range(args.steps):
        optim.zeroing_optimizer(opt, od, od['speaker'])
        fwd_bwd(model, opt, batch_of(opt, rank, s))
        # share_embed: the speaker's backward still adds the shared table's gradient into the listener's segment, so the listener's
        # bucket leaves after the whole backward pass (optimizer.overlap_gradient_exchange); the logit bucket leaves early either way
        want_lst = set() if args.share_embed else {'all'}
        assert set(agents['listener']._pending) == want_lst and set(agents['speaker']._pending) == {'logit'}, \
            'the early buckets did not leave from inside backward()'
        for o in agents.values():
            o.all_reduce_grads()                             # what step() does first; idempotent within a step
        dp_grads.append({a: o.flat.grad.clone() for a, o in agents.items()})
        optim.update_optimizer(od, od['speaker'], opt)
        dp_params.append({a: o.flat.flat.clone() for a, o in agents.items()})
    torch.cuda.synchronize()
    # (1) replicas bit-identical
    for a in agents:
        ref = dp_params[-1][a].clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, dp_params[-1][a]), f'rank {rank}: {a} replica differs from rank 0'

    # ---- single-rank run over all micro-batches (every rank does it; rank 0 reports)
    opt2, model2, od2 = make()
    noises = [NoiseSource(1000 + r) for r in range(world)]
    agents2 = od2['speaker']
    worst_g, worst_p = 0.0, 0.0
    for s in range(args.steps):
        optim.zeroing_optimizer(opt2, od2, od2['speaker'])
        for r in range(world):
            model2.caption_generator.noise = noises[r]
            fwd_bwd(model2, opt2, batch_of(opt2, r, s))
        for a, o in agents2.items():
            o.grad_scale = 1.0 / world
            names = dict(zip(o.flat.names, zip(o.flat.offsets, o.flat.params)))
            for n, (off, p) in names.items():
                if n.endswith('alpha_net.bias'):
                    continue        # a softmax shift: gradient mathematically 0, rounding noise on both sides
                if off is None:
                    continue        # share_embed: the table's gradient lives in (and is compared through) its owner's buffer
                want = o.flat.grad[off:off + p.numel()].double()
                got = dp_grads[s][a][off:off + p.numel()].double()
                err = float((got - want).norm() / (want.norm() + 1e-30))
                worst_g = max(worst_g, err)
                assert err < 1e-4, (s, a, n, err)
        optim.update_optimizer(od2, od2['speaker'], opt2)
        for a, o in agents2.items():
            d = float((o.flat.flat - dp_params[s][a]).abs().max())
            worst_p = max(worst_p, d)
            assert d < 2e-5, (s, a, d)                       # lr 5e-4: a flipped update would show as 1e-3
    torch.cuda.synchronize()
    ok = torch.tensor([1.0], device=dev)
    dist.all_reduce(ok)
    if rank == 0:
        print(json.dumps(dict(dp_equivalence='ok', world=world, backend=args.backend, steps=args.steps,
                              batch_per_rank=args.batch, widths='small' if args.small else 'flagship',
                              max_grad_rel_err=worst_g, max_param_abs_diff=worst_p,
                              buckets={a: {k: list(v) for k, v in o.buckets().items()} for a, o in agents.items()})),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
