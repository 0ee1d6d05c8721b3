"""Prints the kernel / copy timeline of two consecutive bench steps from a rocprofv3 --kernel-trace
--memory-copy-trace CSV directory (dev tool).  usage: python tools/step_timeline.py <dir>"""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"][:24]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_sweep_" in r[2]]
i0, i1 = idx[-8], idx[-6]
t0 = rows[i0 - 3][0]
prev_end = None
for r in rows[i0 - 3:i1 + 1]:
    gap = (r[0] - prev_end) / 1e3 if prev_end else 0.0
    print("%9.2f  gap %6.2f  dur %7.2f  %s" % ((r[0] - t0) / 1e3, gap, (r[1] - r[0]) / 1e3, r[2]))
    prev_end = r[1]
per = (rows[idx[-6]][0] - rows[idx[-16]][0]) / 10e3
print("step period over 10 steps: %.2f us" % per)
