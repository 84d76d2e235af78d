#!/usr/bin/env python3
"""The RCCL branch of the gradient exchange on ONE GPU: a process group of one rank over backend 'nccl' (= RCCL) with
FlatAdam.force_exchange on, so that REAL joint steps run the bucketed all-reduces as a multi-GPU run does - RCCL's
communicator stream, the listener / logit buckets leaving from inside backward(), three asynchronous handles in flight at
update time, the stream-ordered waits in front of the clamp+Adam kernels.  The sum over one rank is the identity, so the
weights after every step must equal those of the same steps without a process group (same batches, same noise stream;
gradient products sum partial tiles with float atomics, so "equal" is to the tolerance of two runs of the same step).

  python tools/rccl_world1.py [--small] [--batch 32] [--steps 3] [--compare 5]

--compare n (the soak form, e.g. --batch 128 --steps 300 --compare 5): float-atomic summation order makes two runs of the SAME
steps drift apart after ~10 steps, so only the first n steps are compared with the run without a group (to the run-to-run
spread); every step of the long run must have a finite loss, the weights must end finite and the sticky status word
(status.py: a hand-off of a one-launch recurrence timed out) must stay clear - three asynchronous exchanges in flight every
step, 300 times over.

Exit code 0 and one JSON line when the equality holds.  (tests/test_gpu_rccl_world1.py runs it in a child process.)"""
import argparse
import contextlib
import json
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--compare', type=int, default=0, help='compare only the first n steps with the plain run (0 = all steps)')
    args = ap.parse_args()
    n_cmp = args.compare if args.compare > 0 else args.steps
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic, status
    from cooperativeimagecaptioning_amd.misc import rewards
    from cooperativeimagecaptioning_amd.noise import NoiseSource
    kw = dict(batch_size=args.batch)
    if args.small:
        kw.update(vocab_size=199, rnn_size=64, input_encoding_size=64, att_hid_size=64, fc_feat_size=128,
                  att_feat_size=128, vse_embed_size=128)
    K = 9 if args.small else 36
    rewards.init_scorer('corpus')

    def run(exchange, steps):
        opt = synthetic.default_opt(**kw)
        torch.manual_seed(0)
        m = models.AlternatingJointModel(opt)
        m.caption_generator.logit.bias.data[0] = 1.0
        m.to(dev).train()
        m.caption_generator.noise = NoiseSource(1000)
        with contextlib.redirect_stdout(sys.stderr):
            od = optim.load_optimizer(m, opt)
        agents = od['speaker']
        if exchange:
            optim.overlap_gradient_exchange(m, od)
        in_flight, params, losses = [], [], []
        pool = [synthetic.make_batch(opt, K=K, seed=1234 + 17 * s, device=dev) for s in range(min(steps, 8))]
        for s in range(steps):
            b = pool[s % len(pool)]
            optim.zeroing_optimizer(opt, od, od['speaker'])
            loss = m(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True,
                     alternating_turn='speaker')
            loss.backward()
            if exchange:
                # the early buckets left from inside backward() and are still handles (RCCL is asynchronous)
                assert set(agents['listener']._pending) == {'all'} and set(agents['speaker']._pending) == {'logit'}, \
                    (agents['listener']._pending, agents['speaker']._pending)
                # update_optimizer() starts 'rest' before the first wait: count what is in flight at that moment
                orig = optim.FlatAdam.all_reduce_grads
                seen = []

                def spy(self, _orig=orig, _seen=seen):
                    _seen.append(sum(len(o._pending) for o in agents.values()))
                    return _orig(self)
                optim.FlatAdam.all_reduce_grads = spy
                try:
                    optim.update_optimizer(od, od['speaker'], opt)
                finally:
                    optim.FlatAdam.all_reduce_grads = orig
                in_flight.append(seen[0])
            else:
                optim.update_optimizer(od, od['speaker'], opt)
            losses.append(loss.detach())
            if s < n_cmp:
                params.append({a: o.flat.flat.clone() for a, o in agents.items()})
        torch.cuda.synchronize()
        losses = [float(x) for x in losses]
        finite = all(x == x and abs(x) != float('inf') for x in losses) and \
            all(bool(torch.isfinite(o.flat.flat).all()) for o in agents.values())
        return losses, params, in_flight, finite

    base_losses, base, _, _ = run(False, n_cmp)
    again_losses, again, _, _ = run(False, n_cmp)            # run-to-run spread of the same steps (float atomics)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    optim.FlatAdam.force_exchange = True
    dp_losses, dp, in_flight, finite = run(True, args.steps)
    optim.FlatAdam.force_exchange = False
    word = int(status.word(dev)[0].item())
    worst, spread = 0.0, 0.0
    for s in range(n_cmp):
        for a in base[s]:
            worst = max(worst, float((dp[s][a] - base[s][a]).abs().max()))
            spread = max(spread, float((again[s][a] - base[s][a]).abs().max()))
    ok = worst <= max(2e-6, 2 * spread) and all(n == 3 for n in in_flight) and finite and word == 0
    for lb, la, ld in zip(base_losses, again_losses, dp_losses):
        ok = ok and abs(lb - ld) <= max(1e-5 * max(1.0, abs(lb)), 2 * abs(lb - la))
    print(json.dumps(dict(rccl_world1='ok' if ok else 'MISMATCH', backend=dist.get_backend(), steps=args.steps,
                          steps_compared=n_cmp, batch=args.batch, widths='small' if args.small else 'flagship',
                          max_param_abs_diff=worst, run_to_run_spread=spread,
                          exchanges_in_flight_at_update=sorted(set(in_flight)), steps_with_three_in_flight=sum(n == 3 for n in in_flight),
                          all_finite=finite, status_word=hex(word),
                          bit_equal=worst == 0.0, losses=dp_losses[:n_cmp], last_loss=dp_losses[-1],
                          losses_without_group=base_losses)), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
