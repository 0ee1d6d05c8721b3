"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_r03/) into the summaries committed under
profiles/ (round 3): kernel stats, the PMC traffic of the sweep kernel per workload, the bench lines."""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r03")
DST = os.path.join(ROOT, "profiles")


def first(pattern):
    # gpurun merges every call's files into gpurun_out/ and never deletes: the newest match is this collection's
    return max(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)


def line(path):
    return json.loads([ln for ln in open(path) if ln.startswith("{")][0])


shutil.copy(first("bench_stats/*/*_kernel_stats.csv"), os.path.join(DST, "r03_kernel_stats_bench.csv"))
shutil.copy(first("est_stats/*/*_kernel_stats.csv"), os.path.join(DST, "r03_kernel_stats_estimate.csv"))
for name in ("bench_plain", "bench_under_rocprof", "estimate_under_rocprof"):
    src = os.path.join(SRC, name + ".json")
    if name == "estimate_under_rocprof":
        shutil.copy(src, os.path.join(DST, "r03_" + name + ".json"))
    else:
        json.dump(line(src), open(os.path.join(DST, "r03_" + name + ".json"), "w"), indent=1)

KEEP = ("k_sweep_q64", "k_finish_coo")
per = {}
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(first("pmc_%s/*/*_counter_collection.csv" % which))))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in KEEP)]
    with open(os.path.join(DST, "r03_pmc_%s_counter_collection.csv" % which), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    agg = collections.defaultdict(list)
    for r in keep:
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    per[counter] = {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

import subprocess
kernel_hash = open(os.path.join(SRC, "kernel_hash.txt")).read().strip()
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "?"
bench = line(os.path.join(SRC, "bench_pmc_fetch.json"))
for tag, leg, key in (("", bench, "k_sweep_q64<12"), ("_c3", bench["configs3"], "k_sweep_q64<13")):
    lay = leg["roofline"]["layout"]
    fk = [k for k in per["FETCH_SIZE"] if key in k][0]
    fetch_kib, nf = per["FETCH_SIZE"][fk]
    write_kib, nw = per["WRITE_SIZE"][fk]
    rd, wr = fetch_kib * 1024 * 2.0, write_kib * 1024
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline (two separate passes, tools/collect_profiles.sh)",
        "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 for 16 B/lane loads per MI355X_MICROARCH.md; non-temporal stream",
        "kernel": "k_sweep_q64",
        "kernel_hash": kernel_hash, "collected_at_commit": commit,
        "workload": {"rows_per_gpu": leg["config"]["rows_per_gpu"], "features": leg["config"]["features"], "nnz": leg["config"]["nnz_per_gpu"],
                     "format": lay["format"], "entry_bytes": 3 + lay["value_bytes"]},
        "units": "FETCH_SIZE / WRITE_SIZE are KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read of 16 B per lane (global_load and buffer_load alike) -> x2; WRITE_SIZE is exact for the piece vectors' streaming stores.  The counters sit on the fabric side of the L2.",
        "kernels": {"k_sweep_q64": {"name": fk, "FETCH_SIZE_KiB_avg": fetch_kib, "launches_fetch": nf, "WRITE_SIZE_KiB_avg": write_kib,
                                    "launches_write": nw, "read_bytes_corrected": rd, "write_bytes": wr,
                                    "hbm_traffic_bytes_per_launch": rd + wr,
                                    "algorithmic_bytes_per_launch": leg["roofline"]["algorithmic_bytes_per_launch"],
                                    "traffic_over_algorithmic": (rd + wr) / leg["roofline"]["algorithmic_bytes_per_launch"]}},
    }
    json.dump(out, open(os.path.join(DST, "r03_pmc_traffic%s.json" % tag), "w"), indent=1)
    print(tag or "c2", out["kernels"]["k_sweep_q64"])
