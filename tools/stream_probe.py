#!/usr/bin/env python3
"""What bounds the per-timestep kernels that stream weights (fused GRU step, BPTT dX products, walkers): the rate ONE CU can
pull, or the bytes ALL CUs pull together?  Workgroups of 512 threads stream their share of a weight-like buffer (regions
shared by the workgroups that would read the same weight tile on one XCD); the same TOTAL tile coverage is timed with 256
workgroups and with 128 bigger ones."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402
import torch
from cooperativeimagecaptioning_amd import _lib
from cooperativeimagecaptioning_amd._lib import lib

lib.cic_debug_stream_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_double), C.c_void_p]


def probe(buf, sink, region_kb, regions, wgs, kb_per_wg, share):
    us = C.c_double(0.0)
    _lib.check(lib.cic_debug_stream_probe(buf.data_ptr(), region_kb * 256, regions, wgs, kb_per_wg * 256, share,
                                          sink.data_ptr(), 200, C.byref(us), torch.cuda.current_stream().cuda_stream), 'probe')
    tot = wgs * kb_per_wg / 1024
    print(f'{wgs:4d} workgroups x {kb_per_wg:4d} KB (regions of {region_kb:4d} KB shared by {share}): {us.value:6.2f} us  '
          f'{tot:6.1f} MB -> {tot / us.value * 1e-3 * 1e3:5.2f} TB/s   {kb_per_wg * 1024 / (us.value * 2100):5.1f} B/clk per workgroup at 2.1 GHz',
          flush=True)


def main():
    buf = torch.randn(64 * 1024 * 1024, device='cuda')          # 256 MB
    sink = torch.zeros(4096, device='cuda')
    print('private regions (every workgroup its own bytes, L2 misses go to the Infinity Cache):')
    for wgs, kb in ((256, 128), (256, 320), (256, 448), (128, 448), (128, 640), (128, 896), (64, 896)):
        probe(buf, sink, kb, 4096 * 64 // kb // 4, wgs, kb, 1)
    print('regions shared by 4 workgroups of one XCD (the fused GRU step: 4 row strips per weight tile):')
    for wgs, kb in ((256, 320), (128, 448), (128, 640)):
        probe(buf, sink, kb, 1024, wgs, kb, 4 if wgs == 256 else 2)


if __name__ == '__main__':
    main()
