#!/usr/bin/env python3
"""Per-kernel time per training step from a rocprofv3 --kernel-trace CSV of bench.py: the trace is cut at the
clamp_adam launches (two per step), the last `steps` steps are kept (warm-up and the roofline timing loop after the
timed region are left out), and the attention kernel is reported separately for the in-step launches.

usage: trace_summary.py <kernel_trace.csv> <steps> [out.md] [sequence.txt] [clamp_adam launches per step: 2]
sequence.txt: every launch of the last step in order (start offset, duration, gap since the previous kernel ended)."""
import collections
import csv
import sys


def short(name):
    for junk in ('void ', '(anonymous namespace)::'):
        name = name.replace(junk, '')
    return name.split('(')[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2])
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
                 int(r['Grid_Size_X']) if 'Grid_Size_X' in r else 0) for r in rows)
    adam = [i for i, e in enumerate(ev) if e[2].startswith('clamp_adam')]
    per = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # (one agent trained: 1)
    s0, s1 = adam[-per * steps - 1] + 1, adam[-1] + 1
    seg = ev[s0:s1]
    busy = sum(e[1] - e[0] for e in seg)
    span = seg[-1][1] - seg[0][0]
    d = collections.defaultdict(lambda: [0, 0])
    for e in seg:
        d[(e[2], e[3])][0] += 1
        d[(e[2], e[3])][1] += e[1] - e[0]
    lines = [f'steps analysed: {steps}; kernels/step {len(seg) / steps:.1f}; GPU busy {busy / steps / 1e6:.3f} ms/step; '
             f'span {span / steps / 1e6:.3f} ms/step (idle {100 * (1 - busy / span):.1f} %)', '',
             '| kernel | grid (threads) | launches/step | avg us | ms/step | % of busy |', '|---|---|---|---|---|---|']
    for (k, gx), (n, t) in sorted(d.items(), key=lambda kv: -kv[1][1]):
        if t / busy < 0.002:
            continue
        lines.append(f'| `{k}` | {gx} | {n / steps:.1f} | {t / n / 1e3:.1f} | {t / steps / 1e6:.3f} | {100 * t / busy:.1f} |')
    text = '\n'.join(lines)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(text + '\n')
    if len(sys.argv) > 4:
        last = ev[adam[-per - 1] + 1:adam[-1] + 1]
        with open(sys.argv[4], 'w') as f:
            prev = last[0][0]
            for e in last:
                f.write(f'{(e[0] - last[0][0]) / 1e3:9.1f} us  dur {(e[1] - e[0]) / 1e3:7.1f}  gap {(e[0] - prev) / 1e3:6.1f}  grid {e[3]:8d}  {e[2][:110]}\n')
                prev = e[1]


if __name__ == '__main__':
    main()
