set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3est
mkdir -p $O
cd $R
KLR_CD_DEBUG=1 python3 tools/estimate_timeline.py > $O/dbg.json 2> $O/dbg.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/estimate_timeline.py > $O/under.json 2> $O/trace.err
f=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_view.py $f agg > $O/agg.txt
python3 tools/trace_view.py $f tl k_fast_rank_out 60 400 > $O/tl.txt
rm -rf $O/trace
tail -2 $O/dbg.json
