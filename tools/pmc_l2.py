#!/usr/bin/env python3
"""L2 behaviour per kernel from ONE rocprofv3 PMC pass of bench.py
(`--pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --kernel-trace`, counters only): requests that reach the eight
L2s per launch (a request = one 128-byte line), their hit rate, and the read requests the L2s pass on to the fabric (Infinity Cache /
HBM; 64 bytes each unless the 32-byte form is used - reported as a count).  What the decode walkers stream per launch and how much of it
comes from beyond the L2.

usage: pmc_l2.py <counter_collection.csv> <out.json>"""
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
    out = {'units': 'per launch averages; a request = one 128-byte line at the L2; ea_rdreq = read requests the L2s send on to the fabric; '
                    'us = GRBM_GUI_ACTIVE / 8 XCDs / 2400 MHz (the counter pass slows short kernels: compare ratios, not times)',
           'kernels': []}
    for (name, grid), a in sorted(agg.items(), key=lambda kv: -(kv[1].get('TCC_HIT_sum', 0.0) + kv[1].get('TCC_MISS_sum', 0.0))):
        n = a['n']
        hit, miss, ea = a.get('TCC_HIT_sum', 0.0) / n, a.get('TCC_MISS_sum', 0.0) / n, a.get('TCC_EA0_RDREQ_sum', 0.0) / n
        if hit + miss <= 0:
            continue
        out['kernels'].append({'kernel': name, 'grid_threads': grid, 'launches': int(n), 'l2_requests': round(hit + miss),
                               'l2_request_MB': (hit + miss) * 128 / 1e6, 'hit_rate': hit / (hit + miss), 'l2_miss_MB': miss * 128 / 1e6,
                               'ea_rdreq': round(ea), 'ea_rd_MB_at_64B': ea * 64 / 1e6,
                               'us': a.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / n / 2400.0})
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for k in out['kernels'][:16]:
        print({a: (round(b, 3) if isinstance(b, float) else b) for a, b in k.items()})


if __name__ == '__main__':
    main()
