"""Tools that flip dispatch switches or read in-kernel stamps need the DEVELOPMENT build of the library
(libcic_hip_dev.so: the same sources with -DCIC_DEVTOOLS, include/cic_dev.h).  Import this module before anything of
cooperativeimagecaptioning_amd: it builds that library if needed and points the package's loader at it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
assert 'cooperativeimagecaptioning_amd._lib' not in sys.modules, 'import tools/_devlib.py before the package'
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('_cic_build', os.path.join(ROOT, 'cooperativeimagecaptioning_amd', 'build.py'))
_build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_build)
os.environ['CIC_HIP_LIB'] = _build.build(dev=True, verbose=False)
