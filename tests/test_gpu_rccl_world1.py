"""GPU: the RCCL branch of the bucketed gradient exchange on the one GPU of the box: a process
group of ONE rank over backend 'nccl' with the exchange forced on - RCCL's stream, the listener and logit buckets leaving
from inside backward(), three asynchronous all-reduces in flight at update time - on real joint steps; weights after every
step equal those of the same steps without a process group (tools/rccl_world1.py asserts it and prints a JSON line)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize('extra', [['--small', '--batch', '8', '--steps', '3'], ['--batch', '32', '--steps', '2'],
                                   # the headline shape, 300 steps with three exchanges in flight each: the first
                                   # 5 steps equal to the run-to-run spread, every step finite, the status word clear
                                   ['--batch', '128', '--steps', '300', '--compare', '5']])
def test_forced_exchange_over_rccl_in_a_group_of_one_equals_the_plain_step(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_world1.py')] + extra, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=570)
    assert r.returncode == 0, r.stdout[-2000:] + '\n' + r.stderr[-4000:]
    doc = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert doc['rccl_world1'] == 'ok' and doc['backend'] == 'nccl'
    assert doc['exchanges_in_flight_at_update'] == [3]          # listener all + speaker logit + speaker rest, every step
    assert doc['steps_with_three_in_flight'] == doc['steps'] and doc['all_finite'] and doc['status_word'] == '0x0'
