#!/bin/bash
# Dev tool (GPU box, through gpurun): where k_gram_offdiag<true> spends its time -- LDS conflicts / instruction mix / busy cycles.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gram_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/tools/estimate_timeline.py > /dev/null 2> $OUT/$tag.err
  echo "$set rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gram_offdiag" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(acc.items()):
    print("%-24s %14.0f per launch (%d launches)" % (k, v[0] / max(v[1], 1), v[1]))
PY
