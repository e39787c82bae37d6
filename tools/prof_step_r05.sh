R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f2; mkdir -p $O; cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 2 --cpu-baseline off --verify off --traffic off --f32-split off --conv-log $O/conv_log.json > $O/kt.log 2>&1
cd $R
grep '^{' $O/kt.log > $O/bench_n1_quick.json
python tools/step_breakdown.py $(find $O/kt -name "*kernel_trace.csv") 3 $O/conv_log.json > $O/step_breakdown.md 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.csv" -size +3M -delete
head -16 $O/step_breakdown.md; grep "all skp_wino4" $O/step_breakdown.md
for i in 1 2; do python bench.py --steps 20 --warmup 3 --cpu-baseline off --verify off --traffic off --kernel-iters 1 --f32-split off 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench', round(d['value'],3), round(d['ms_per_step'],3))"; done
