#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of bench.py (configs[2] and configs[3]'s shard), separate PMC passes for FETCH_SIZE and WRITE_SIZE,
#   kernel stats of three whole --lambda-auto=100 estimates.  Raw output under gpurun_out/; tools/reduce_profiles.py
#   turns it into the summaries committed under profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.err
echo "stats rc=$?" >> $OUT/log.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch rc=$?" >> $OUT/log.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "write rc=$?" >> $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/est_stats -- python3 $R/tools/estimate_timeline.py > $OUT/estimate_under_rocprof.json 2> $OUT/est_stats.err
echo "est rc=$?" >> $OUT/log.txt
cd $R && python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain rc=$?" >> $OUT/log.txt
find $OUT -name "*.csv" | head -40 >> $OUT/log.txt
python3 -c "import sys; sys.path.insert(0, \"$R\"); import kmerlr_amd as K; print(K.kernel_hash())" > $OUT/kernel_hash.txt
cat $OUT/log.txt
