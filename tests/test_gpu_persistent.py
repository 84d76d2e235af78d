"""GPU: the in-launch / no-host-sync control structures of this round, each checked against the plain form it replaces
(the development build flips between them; tools/ print a JSON line and exit non-zero on a mismatch):
  * gru_seq_kernel - the listener's whole GRU pass as ONE launch with W_hh stationary in registers and per-strip
    hand-offs of the hidden state inside the launch - must reproduce the one-launch-per-step kernel BIT FOR BIT, also with
    another stream saturating the chip while it runs; its backward twin gru_seq_bwd_kernel (the whole GRU BPTT loop in one
    launch) must reproduce the gradients of the launch-per-step loop to summation-order tolerance (tools/gru_seq_check.py);
  * spk_bptt_seq_kernel - the speaker's BPTT loop (cell, a2c product, attention, h2h + h2att product; three hand-offs per
    step) as ONE launch - same tokens and loss, every gradient within 1e-5 of its largest element of the four-launches-per-
    step loop, with full-length captions, captions that end early and ragged region masks (tools/bptt_seq_check.py);
  * spk_teacher_seq_kernel - the teacher-forced recurrence of AttModel.forward (BASELINE configs[1]) as ONE launch - loss and
    gradients of the MLE step against the three-launches-per-step form (tools/teacher_seq_check.py);
  * the device-side early stop of the decode loops (AttModel.py:401-408) must leave tokens, lengths, loss and gradients
    of a full-width joint step exactly as the full loops give them (tools/early_stop_check.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool)] + list(args), cwd=ROOT, capture_output=True, text=True,
                       timeout=560)
    assert r.returncode == 0, r.stdout[-2000:] + '\n' + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])


@pytest.mark.timeout(600)
def test_one_launch_gru_pass_equals_the_per_step_kernel_bit_for_bit():
    doc = _run('gru_seq_check.py', '--iters', '20')
    assert doc['gru_seq_check'] == 'ok' and doc['uneven_load']['mismatches'] == 0 and doc['uneven_load']['backward_mismatches'] == 0
    for mode in ('generated', 'labels'):
        assert all(doc[mode]['outputs_bit_equal'].values()) and doc[mode]['finite']


@pytest.mark.timeout(600)
def test_early_stop_of_the_decode_loops_changes_nothing_observable():
    doc = _run('early_stop_check.py', '--iters', '5')
    assert doc['early_stop_check'] == 'ok' and max(doc['L_sampled'], doc['L_greedy']) < 16


@pytest.mark.timeout(600)
def test_one_launch_speaker_bptt_loop_equals_the_per_step_loop():
    doc = _run('bptt_seq_check.py', '--iters', '5')
    assert doc['bptt_seq_check'] == 'ok' and doc['uneven_load']['mismatches'] == 0
    for case in ('full_length', 'early_end', 'ragged_regions'):
        assert doc[case]['forward_equal'] and doc[case]['finite'] and doc[case]['max_grad_rel_diff'] < 1e-5
    assert doc['early_end']['L'] < 16


@pytest.mark.timeout(600)
def test_one_launch_teacher_forced_recurrence_equals_the_per_step_launches():
    doc = _run('teacher_seq_check.py', '--iters', '5')
    assert doc['teacher_seq_check'] == 'ok'
    for B in (64, 128):
        assert doc[f'B{B}_uneven_load']['mismatches'] == 0
        for case in ('plain', 'ragged_regions'):
            d = doc[f'B{B}_{case}']
            assert d['finite'] and d['loss_rel_diff'] < 2e-6 and d['max_grad_rel_diff'] < 1e-5 and d['n_grad_buffers'] > 0


@pytest.mark.timeout(900)
def test_one_launch_loops_at_a_batch_that_is_not_a_multiple_of_their_16_row_strips():
    doc = _run('gru_seq_check.py', '--iters', '2', '--batch', '100')
    assert doc['gru_seq_check'] == 'ok' and doc['uneven_load']['mismatches'] == 0 and doc['uneven_load']['backward_mismatches'] == 0
    doc = _run('bptt_seq_check.py', '--iters', '2', '--batch', '100')
    assert doc['bptt_seq_check'] == 'ok' and doc['uneven_load']['mismatches'] == 0


@pytest.mark.timeout(1200)
def test_one_launch_loops_walk_a_batch_of_more_than_128_rows_in_row_blocks():
    """256 rows = two full row blocks of 8 strips (configs[3]); 144 rows = a full block and a block of one strip."""
    for B in ('256', '144'):
        doc = _run('gru_seq_check.py', '--iters', '2', '--batch', B)
        assert doc['gru_seq_check'] == 'ok' and doc['uneven_load']['mismatches'] == 0 and doc['uneven_load']['backward_mismatches'] == 0
        doc = _run('bptt_seq_check.py', '--iters', '2', '--batch', B)
        assert doc['bptt_seq_check'] == 'ok' and doc['uneven_load']['mismatches'] == 0
    doc = _run('teacher_seq_check.py', '--iters', '2', '--batches', '256,144')
    assert doc['teacher_seq_check'] == 'ok'
    for B in (256, 144):
        assert doc[f'B{B}_uneven_load']['mismatches'] == 0
