R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default.json; echo default $?
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 2 --cpu-baseline off --verify off --traffic off --f32-split off --conv-log $O/conv_log.json > $O/kt.log 2>&1
cd $R
grep '^{' $O/kt.log > $O/bench_n1_quick.json
python tools/step_breakdown.py $(find $O/kt -name "*kernel_trace.csv") 3 $O/conv_log.json > $O/step_breakdown.md 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.csv" -size +3M -delete
for cfg in "--images-per-rank 1" "--images-per-rank 2" "--tokens 500" "--scaling strong --global-batch 8" "--model sd21 --top-k 30 --candidates 50" "--model sdxl --images-per-rank 2"; do n=$(echo $cfg | tr -d " -"); python bench.py $cfg --steps 10 --warmup 3 --cpu-baseline off --verify off --traffic off > $O/cfg_$n.log 2>&1; grep '^{' $O/cfg_$n.log > $O/cfg_$n.json; python -c "
import json; d=json.load(open('$O/cfg_$n.json')); print('$n', round(d['value'],3), round(d['ms_per_step'],2), (d.get('f32_split') or {}).get('value'))"; done
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['conv_all_launches']['frac'], d['f32_split'].get('value'), d['cpu_baseline']['value'])"
head -16 $O/step_breakdown.md
