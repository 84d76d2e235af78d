"""GPU: failures of the one-launch recurrences are LOUD and NON-DESTRUCTIVE ("Next round" item 1).

* the fused clamp+Adam kernel clamps like torch's clamp_ (misc/utils.py:65-69): a NaN gradient stays NaN and reaches the
  parameter, as in the reference and the oracle - round 3's fminf(fmaxf(NaN, -c), c) turned it into a finite -c update;
* the guarded form (cic_clamp_adam_guarded) leaves parameters, moments and gradient untouched while the sticky status word
  is set, and marks CIC_STATUS_UPDATE_SKIPPED;
* train.LossLog carries the word to the host inside the loss copy and raises CicError naming the loop and the iteration;
* development build (tools/handoff_fault_check.py): a forced hand-off time-out in each of the four loops is reported, the
  step's update is skipped (weights bit-equal to before), the launch ends within a few spin bounds, and the next step after
  clearing the word is clean; a second-stream kernel that HOLDS 8 / 32 CUs for 2 ms beside the loops (the stand-in for a
  collective) causes no failure and no difference in results - its slowdown is recorded (profiles/r04_handoff_fault_check.json).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _adam_inputs(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g) * 0.3            # a good share beyond the +-0.1 clamp
    m = torch.randn(n, generator=g) * 0.01
    v = torch.rand(n, generator=g) * 1e-3
    return p, grad, m, v


def test_clamp_adam_propagates_nan_like_torch_clamp_and_the_oracle():
    from cooperativeimagecaptioning_amd import engine
    from oracle import joint as J
    n = 4099                                            # not a multiple of 4: the scalar tail of the kernel too
    p, grad, m, v = _adam_inputs(n)
    bad = [0, 5, 1027, n - 1]
    for i in bad:
        grad[i] = float('nan')
    grad[7], grad[8] = float('inf'), float('-inf')      # clamp_ maps them to +-clip, so does the kernel
    # oracle: clamp (torch.clamp propagates NaN) + Adam, step 3 with given moments
    st = {'w': dict(step=2, exp_avg=m.clone(), exp_avg_sq=v.clone())}
    P = {'w': p.clone()}
    J.clamp_adam_step(P, {'w': grad.clone()}, st, 5e-4, 0.1)
    dp, dg, dm, dv = (t.cuda() for t in (p, grad, m, v))
    engine.clamp_adam(dp, dg, dm, dv, 5e-4, 3, 0.1, guarded=False)
    torch.cuda.synchronize()
    out = dp.cpu()
    assert torch.isnan(out[bad]).all() and torch.isnan(P['w'][bad]).all()
    good = torch.ones(n, dtype=torch.bool)
    good[bad] = False
    assert torch.isfinite(out[good]).all()
    np.testing.assert_allclose(out[good].numpy(), P['w'][good].numpy(), rtol=2e-6, atol=2e-7)
    assert torch.isnan(dm.cpu()[bad]).all() and torch.isnan(dv.cpu()[bad]).all()


def test_guarded_clamp_adam_skips_the_update_while_the_status_word_is_set():
    from cooperativeimagecaptioning_amd import engine, status
    n = 1 << 16
    p, grad, m, v = (t.cuda() for t in _adam_inputs(n, 1))
    ref = [t.clone() for t in (p, grad, m, v)]
    status.clear()
    try:
        status.word()[0] = status.BPTT
        engine.clamp_adam(p, grad, m, v, 5e-4, 1, 0.1, zero_grad=True)
        torch.cuda.synchronize()
        for a, b in zip((p, grad, m, v), ref):
            assert torch.equal(a, b)                    # nothing touched: not even the gradient is cleared
        assert int(status.word()[0]) == status.BPTT | status.UPDATE_SKIPPED
        with pytest.raises(Exception) as e:
            status.check()
        assert 'spk_bptt_seq_kernel' in str(e.value) and 'skipped' in str(e.value)
        status.clear()
        engine.clamp_adam(p, grad, m, v, 5e-4, 1, 0.1, zero_grad=True)
        torch.cuda.synchronize()
        assert not torch.equal(p, ref[0]) and float(grad.abs().max()) == 0.0 and int(status.word()[0]) == 0
    finally:
        status.clear()


def test_losslog_carries_the_status_word_and_raises_naming_loop_and_iteration():
    from cooperativeimagecaptioning_amd import status, train as T, _lib
    dev = torch.device('cuda', 0)
    status.clear(dev)
    try:
        log = T.LossLog(dev)
        loss = torch.tensor(1.25, device=dev)
        seen = []
        log.push(dict(iteration=7, epoch=0, turn='speaker', host_s=0.0, to_history=False), loss, {'a': torch.tensor(2.0, device=dev)})
        log.pop(lambda meta, l, terms: seen.append((meta['iteration'], l, terms)), block_to=0)
        assert seen == [(7, 1.25, {'a': 2.0})]            # a clear word costs nothing and changes nothing
        status.word(dev)[0] = status.GRU_BWD | status.UPDATE_SKIPPED
        log.push(dict(iteration=8, epoch=0, turn='speaker', host_s=0.0, to_history=False), loss, {'a': torch.tensor(2.0, device=dev)})
        with pytest.raises(_lib.CicError) as e:
            log.pop(lambda *a: seen.append(a), block_to=0)
        assert 'gru_seq_bwd_kernel' in str(e.value) and 'iteration 8' in str(e.value) and 'CIC_SHARED_DEVICE' in str(e.value)
        assert len(seen) == 1                             # the poisoned line was not emitted
    finally:
        status.clear(dev)


@pytest.mark.timeout(900)
def test_forced_handoff_timeouts_are_reported_and_skip_the_update_and_held_cus_only_cost_time():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'handoff_fault_check.py')], cwd=ROOT, capture_output=True,
                       text=True, timeout=860)
    assert r.returncode == 0, r.stdout[-3000:] + '\n' + r.stderr[-3000:]
    doc = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert doc['handoff_fault_check'] == 'ok'
    for loop in ('gru_fwd', 'gru_bwd', 'bptt', 'teacher'):
        d = doc[loop]
        assert d['weights_and_moments_untouched'] and d['losslog_raised'] and d['names_loop'] and d['next_step_updates']
        assert d['step_wall_s'] < 2.0 and d['clean_after_clear'] and d['next_step_finite']
    for k in ('hold_0_cus', 'hold_8_cus', 'hold_32_cus'):
        assert doc['occupancy'][k]['mismatches'] == 0 and doc['occupancy'][k]['status_word'] == '0x0'
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'r04_handoff_fault_check.json'), 'w') as f:
        json.dump(doc, f, indent=1)
