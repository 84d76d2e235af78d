"""MI355X-native hot path of the cooperative image-captioning joint training step.

Host side mirrors the reference's module API (models.*, misc.rewards, opts, train); the
compute runs in hand-written HIP kernels for gfx950 behind the C ABI of include/cic.h
(libcic_hip.so).  Importing the package loads that library and fails loudly without it.
"""
from . import _lib  # noqa: F401  (raises CicError if libcic_hip.so is missing)

__version__ = '0.1.0'
