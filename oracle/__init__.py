"""CPU oracle for the cooperative-captioning joint step.

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 PyTorch-CPU / numpy
restatement of the reference algorithm (vgilad/CooperativeImageCaptioning) for
the AlternatingJointModel speaker<->listener hot path.  It exists so that the
HIP path can be checked against it; it is never the thing shipped or measured.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package ``cooperativeimagecaptioning_amd`` must
never import anything from here.

Parity pinning: the reference ships no tests or golden vectors,
so every function here is pinned against outputs of the reference itself,
generated in the build container by ``tools/gen_golden.py`` (which imports the
reference unmodified through a small compat harness) and committed as small
fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py`` replays them.

All stochastic operations take their noise as explicit arguments (Gumbel
uniforms, dropout keep-masks, multinomial picks, partial-sampling row masks) so
that the reference, the oracle and the HIP kernels can be driven with identical
randomness.
"""
from . import speaker, listener, ciderd, joint  # noqa: F401
