#!/usr/bin/env python3
"""MFMA utilisation per kernel from ONE rocprofv3 PMC pass of bench.py
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace`, counters only).

  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
                reports the sum over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back") - the share of SIMD-cycles in which
                the matrix pipe was busy;
  mfma_flop   = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 (flops issued on f32 MFMA), so flop / cycles / 1024 SIMDs against the
                64 flop/clk/SIMD of the f32 MFMA is the same utilisation from the instruction side.

usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def short(name):
    for junk in ('void ', '(anonymous namespace)::'):
        name = name.replace(junk, '')
    return name.split('(')[0]


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(sys.argv[1])):
        d = r.get('Dispatch_Id') or r.get('Dispatch_ID') or (r['Kernel_Name'] + r.get('Correlation_Id', ''))
        per[d][r['Counter_Name']] += float(r['Counter_Value'])
        meta[d] = (short(r['Kernel_Name']), int(r['Grid_Size']))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d, c in per.items():
        a = agg[meta[d]]
        a['n'] += 1
        for k, v in c.items():
            a[k] += v
    out = {'units': 'per launch averages; cycles = GRBM_GUI_ACTIVE / 8 XCDs; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs)',
           'kernels': []}
    for (name, grid), a in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)):
        n = a['n']
        cyc = a.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / n
        busy = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n
        flop = a.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) * 512.0 / n
        if busy <= 0:
            continue
        out['kernels'].append({'kernel': name, 'grid_threads': grid, 'launches': int(n), 'cycles': round(cyc),
                               'mfma_busy_cycles': round(busy), 'mfma_util': busy / (cyc * 1024.0) if cyc else None,
                               'mfma_f32_flop': flop,
                               'mfma_util_from_flop': flop / (cyc * 1024.0 * 64.0) if cyc else None})
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for k in out['kernels'][:14]:
        print(k)


if __name__ == '__main__':
    main()
