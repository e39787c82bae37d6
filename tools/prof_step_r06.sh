# usage (on the GPU box): bash tools/prof_step_r06.sh <tag> [bench flags...]   -> gpurun_out/r06/<tag>_{step_breakdown.md,bench.json,kernel_stats.csv}
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/kt_$TAG -o kt -- python $R/bench.py --steps 3 --warmup 2 --cpu-baseline off --verify off --traffic off --f32-split off --conv-log $O/${TAG}_conv_log.json "$@" > $O/${TAG}_kt.log 2>&1
cd $R
grep '^{' $O/${TAG}_kt.log > $O/${TAG}_bench.json
python tools/step_breakdown.py $(find $O/kt_$TAG -name "*kernel_trace.csv") 3 $O/${TAG}_conv_log.json > $O/${TAG}_step_breakdown.md 2>&1
find $O/kt_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
find $O -name "*.csv" -size +3M -delete; rm -rf $O/kt_$TAG
head -16 $O/${TAG}_step_breakdown.md
