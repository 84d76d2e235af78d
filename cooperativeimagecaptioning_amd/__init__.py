"""MI355X-native hot path of the cooperative image-captioning joint training step.

Host side mirrors the reference's module API (models.*, misc.rewards, opts, train); the
compute runs in hand-written HIP kernels for gfx950 behind the C ABI of include/cic.h
(libcic_hip.so).  Every compute module (engine, ops, models, optimizer, ...) imports ``_lib``, which
loads that library and raises CicError without it: there is no fallback path.  The package import
itself stays free of it so that ``cooperativeimagecaptioning_amd.build`` can run on a fresh checkout.
"""
__version__ = '0.1.0'
