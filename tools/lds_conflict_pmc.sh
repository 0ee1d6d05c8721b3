#!/bin/bash
# Dev tool (GPU box, through gpurun): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the sweep kernel with the stream's columns
# in ascending order (KLR_Q64_BANK=0) and in the bank-aware order (default), configs[2].
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lds_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 0 16; do
  export KLR_Q64_BANK=$b
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/bank$b -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_bank$b.json 2> $OUT/bank$b.err
  echo "bank=$b rc=$?"
done
python3 - <<PY
import csv, glob, collections
for b in (0, 16):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/bank%d/**/*counter_collection.csv" % b, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sweep_q64" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print("KLR_Q64_BANK=%d" % b, {k: v[0] / max(v[1], 1) for k, v in acc.items()})
PY
