# Round-6 evidence on ONE MI355X box (gpurun -- bash tools/evidence_r06.sh <tag>): everything lands in gpurun_out/r06/<tag>/
#   bench_default.json         python bench.py (default flags: the driver's command, CPU baseline included)
#   step_breakdown[_ipr1].md   rocprofv3 --kernel-trace of 3 steady-state steps at 4 images / 1 image per rank (tools/step_breakdown.py)
#   bench_ipr{1,2}.json, cfg_*.json   the other per-rank shapes / BASELINE configurations (no CPU leg)
#   pmc_map.json               counter passes over tools/map_bench.py (VALU / LDS counters of the north-star kernels)
R=$GRAFT_REPO_ROOT; TAG=${1:-final}; O=$R/gpurun_out/r06/$TAG; mkdir -p $O; cd $R
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default.json; echo default rc=$?
# boxes differ by ~5 % (sustained clock): with MAXMS set, a box slower than that is not worth the rest of the run
if [ -n "$MAXMS" ]; then python -c "
import json,sys; d=json.load(open('$O/bench_default.json')); print('default ms/step', d['ms_per_step']); sys.exit(0 if d['ms_per_step'] <= float('$MAXMS') else 3)" || exit 3; fi
cd /tmp; export TMPDIR=/tmp
for ipr in 4 1; do
  sfx=""; [ $ipr = 1 ] && sfx="_ipr1"
  rocprofv3 --kernel-trace --stats -f csv -d $O/kt$sfx -o kt -- python $R/bench.py --images-per-rank $ipr --steps 3 --warmup 2 --cpu-baseline off --verify off --traffic off --f32-split off --conv-log $O/conv_log$sfx.json > $O/kt$sfx.log 2>&1
  python $R/tools/step_breakdown.py $(find $O/kt$sfx -name "*kernel_trace.csv") 3 $O/conv_log$sfx.json > $O/step_breakdown$sfx.md 2>&1
  find $O/kt$sfx -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats$sfx.csv \;
  rm -rf $O/kt$sfx
done
cd $R
for cfg in "--images-per-rank 1" "--images-per-rank 2" "--images-per-rank 1 --graph off" "--images-per-rank 2 --graph off" "--tokens 500" "--scaling strong --global-batch 8" "--model sd21 --top-k 30 --candidates 50" "--model sdxl --images-per-rank 2"; do
  n=$(echo $cfg | tr -d " -"); python bench.py $cfg --steps 10 --warmup 3 --cpu-baseline off --verify off --traffic off > $O/cfg_$n.log 2>&1; grep '^{' $O/cfg_$n.log > $O/cfg_$n.json
  python -c "
import json; d=json.load(open('$O/cfg_$n.json')); print('$n', round(d['value'],3), round(d['ms_per_step'],2), (d.get('f32_instr') or {}).get('ms_per_step'), round(d['launch_thread_cpu_ms_per_step'],2), d.get('step_from_idle'), d['captured_step'].get('group_sizes_captured'))"; done
bash tools/prof_kernels.sh map6 tools/map_bench.py --skip-dense --iters 3 > $O/pmc_map.log 2>&1; cp gpurun_out/prof_map6/pmc_summary.json $O/pmc_map.json 2>/dev/null
python tools/rccl_smoke.py > $O/rccl_smoke.json 2>$O/rccl_smoke.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('default', d['value'], d['ms_per_step'], r['frac'], r['launch_us'], r['conv_all_launches']['frac'], d['f32_instr']['ms_per_step'], d['cpu_baseline']['value'], d['roofline_attn_map']['valu']['frac'])"
head -16 $O/step_breakdown.md
