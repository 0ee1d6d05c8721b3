#!/usr/bin/env python3
"""Timeline / per-kernel totals of a rocprofv3 kernel trace csv.
   trace_view.py <kernel_trace.csv> agg|tl [anchor-kernel-substring] [which-occurrence] [count]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mode = sys.argv[2] if len(sys.argv) > 2 else 'agg'
anchor = sys.argv[3] if len(sys.argv) > 3 else None
occ = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cnt = int(sys.argv[5]) if len(sys.argv) > 5 else 80
short = lambda n: re.sub(r'[(].*', '', n).replace('void ', '')[:44]
if anchor:
    idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
    i0 = idx[occ]
else:
    i0 = 0
if mode == 'tl':
    t0 = int(rows[i0]['Start_Timestamp']); prev = t0
    for r in rows[i0:i0 + cnt]:
        s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
        print(f"{(s - t0) / 1000:9.1f} gap {(s - prev) / 1000:6.1f} dur {(e - s) / 1000:7.1f}  {short(r['Kernel_Name'])}")
        prev = e
else:
    sub = rows[i0:]
    agg = collections.defaultdict(lambda: [0, 0]); busy = 0
    for r in sub:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg[short(r['Kernel_Name'])]; a[0] += 1; a[1] += d; busy += d
    span = int(sub[-1]['End_Timestamp']) - int(sub[0]['Start_Timestamp'])
    print(f"span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, {len(sub)} kernels")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
        print(f"{v[1] / 1e3:9.1f} us {v[0]:5d} x {v[1] / v[0] / 1e3:7.1f}  {k}")
