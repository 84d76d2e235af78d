#!/usr/bin/env python3
"""Does throughput-bound leaf work (weight-gradient GEMMs) hide beside the latency-bound BPTT loop when it runs on a
second stream that is confined to a subset of the CUs (hipExtStreamCreateWithCUMask)?

The bench step is run with three leaf-shaped GEMMs (logit dW, GRU dW_hh, GRU dW_ih shapes; results discarded) added at
the point where the speaker's BPTT loop starts (the speaker_logit_grads_ready hook):
  none      no extra work (the plain step)
  serial    the extra GEMMs on the step's own stream
  aux:N     the extra GEMMs on a second stream limited to N CUs (N = 256: no mask), joined before the optimizer
  aux:N,main:M  ... and the step itself on a stream limited to M CUs
What is hidden = serial - aux.   usage: cumask_probe.py none serial aux:256 aux:128 ..."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import torch
from cooperativeimagecaptioning_amd import models, ops, optimizer as optim, synthetic
from cooperativeimagecaptioning_amd.misc import rewards

hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(n_cus, pattern='low'):
    """A stream whose kernels may use n_cus of the 256 CUs.  pattern 'low': mask bits [0, n); 'high': bits [256-n, 256)."""
    if n_cus >= 256:
        return torch.cuda.Stream()
    bits = range(n_cus) if pattern == 'low' else range(256 - n_cus, 256)
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, f'hipExtStreamCreateWithCUMask -> {rc}'
    return torch.cuda.ExternalStream(s.value)


def main():
    specs = sys.argv[1:] or ['none', 'serial', 'aux:256', 'aux:128']
    opt = synthetic.default_opt(batch_size=128)
    torch.manual_seed(0)
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt).cuda().train()
    od = optim.load_optimizer(model, opt)
    o = od['speaker']
    b = synthetic.make_batch(opt, seed=1, device='cuda')
    optim.fuse_zero_grad(od)
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(1)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    leaves = [(R(2048, 9488), R(2048, 512), torch.empty(9488, 512, device=dev)),
              (R(2176, 3072), R(2176, 1024), torch.empty(3072, 1024, device=dev)),
              (R(2176, 3072), R(2176, 512), torch.empty(3072, 512, device=dev))]

    def leaf_work():
        for A, B_, C_ in leaves:
            ops.gemm(A, B_, C_, a_kc=False, b_kc=False, sum_order_free=True)

    state = dict(mode='none', aux=None, ev_fork=torch.cuda.Event(), ev_join=torch.cuda.Event())

    def hook():
        if state['mode'] == 'serial':
            leaf_work()
        elif state['mode'] == 'aux':
            cur = torch.cuda.current_stream()
            state['ev_fork'].record(cur)
            with torch.cuda.stream(state['aux']):
                state['aux'].wait_event(state['ev_fork'])
                leaf_work()
                state['ev_join'].record(state['aux'])
    model.speaker_logit_grads_ready = hook

    def run(n):
        for _ in range(n):
            optim.zeroing_optimizer(opt, od, o)
            loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'], is_alternating=True,
                         alternating_turn='speaker')
            loss.backward()
            if state['mode'] == 'aux':
                torch.cuda.current_stream().wait_event(state['ev_join'])
            optim.update_optimizer(od, o, opt)

    for rep in range(2):
        for spec in specs:
            kv = dict(p.split(':') for p in spec.split(',') if ':' in p)
            state['mode'] = spec.split(',')[0].split(':')[0]
            pat = kv.get('pat', 'low')
            if state['mode'] == 'aux':
                state['aux'] = masked_stream(int(kv['aux']), pat)
            main_s = masked_stream(int(kv['main']), 'high' if pat == 'low' else 'low') if 'main' in kv else torch.cuda.current_stream()
            with torch.cuda.stream(main_s):
                run(5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(40)
                torch.cuda.synchronize()
            print(f'{spec}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step', flush=True)


if __name__ == '__main__':
    main()
