# step timeline of both shapes under a kernel trace (gpurun: bash tools/run_step_trace.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/steptrace
rm -rf $O; mkdir -p $O
for sh in c2 c3; do
  python3 $R/tools/bench_step_trace.py $sh > $O/plain_$sh.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$sh -- python3 $R/tools/bench_step_trace.py $sh 20 > $O/under_$sh.txt 2> $O/err_$sh.txt
  f=$(ls $O/tr_$sh/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_view.py $f tl k_row_pass -12 14 > $O/tl_$sh.txt
  rm -rf $O/tr_$sh
done
cat $O/plain_c2.txt $O/plain_c3.txt
