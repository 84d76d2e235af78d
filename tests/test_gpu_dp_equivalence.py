"""GPU: the data-parallel row of the scope table on REAL joint steps.  Two ranks (gloo, both on
cuda:0: a one-GPU box cannot hold an RCCL pair) each run one micro-batch through the bucketed gradient exchange; the
result must equal ONE rank accumulating both micro-batches with grad_scale = 1/2 (tools/dp_equivalence.py asserts
the equalities and prints a JSON line).  The same script runs under RCCL with one GPU per rank on a multi-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize('extra', [['--small', '--batch', '8', '--steps', '3'], ['--batch', '32', '--steps', '2'],
                                   ['--small', '--batch', '8', '--steps', '3', '--share-embed']])
def test_two_ranks_equal_one_rank_accumulating_two_micro_batches(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tools', 'dp_equivalence.py'), '--backend', 'gloo',
           '--same-device'] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=570)
    assert r.returncode == 0, r.stdout[-2000:] + '\n' + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    doc = json.loads(line)
    assert doc['dp_equivalence'] == 'ok' and doc['world'] == 2
    assert set(doc['buckets']['speaker']) == {'rest', 'logit'} and set(doc['buckets']['listener']) == {'all'}
