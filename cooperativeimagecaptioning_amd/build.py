"""Builds libcic_hip.so (gfx950) in-tree with hipcc.  No torch headers, no CUDA shims.

  python -m cooperativeimagecaptioning_amd.build [--force] [--dev]

--dev builds libcic_hip_dev.so from the same sources with -DCIC_DEVTOOLS: dispatch switches become settable
(include/cic_dev.h) and the kernels carry their phase-stamp code.  Only tools/ load it (tools/_devlib.py)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libcic_hip.so')
LIB_DEV = os.path.join(HERE, 'libcic_hip_dev.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-Wall', '-Wno-unused-function', '-Wno-constant-logical-operand']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))


def needs_build(dev=False):
    lib = LIB_DEV if dev else LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True, dev=False):
    lib = LIB_DEV if dev else LIB
    if not force and not needs_build(dev):
        return lib
    objs = []
    hdr_t = max(os.path.getmtime(h) for h in
                glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h')))
    procs = []
    for src in sources():
        obj = src[:-4] + ('.dev.o' if dev else '.o')
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            continue
        cmd = [HIPCC] + FLAGS + (['-DCIC_DEVTOOLS'] if dev else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + src)
    # link under a temporary name and rename into place: a process that loads the library while another one builds it
    # (ranks of a multi-GPU launch) never maps a half-written file
    tmp = lib + f'.tmp{os.getpid()}'
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, lib)
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv, dev='--dev' in sys.argv)
