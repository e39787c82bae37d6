#!/bin/bash
# rocprofv3 passes over a micro-benchmark: kernel trace + five separate counter groups (never combined with traces).
#   tools/prof_kernels.sh <tag> <python script + args...>     -> gpurun_out/prof_<tag>/{kt,pmc1..pmc5}/ + pmc_summary.json
# e.g. the file bench.py reads `roofline.traffic` from (copy the summary to profiles/rNN_pmc_<what>.json):
#   gpurun -- tools/prof_kernels.sh kb tools/kbench.py --iters 3
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python $ROOT/"$@" > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc1 -o p -- python $ROOT/"$@" > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -f csv -d $OUT/pmc2 -o p -- python $ROOT/"$@" > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY -f csv -d $OUT/pmc3 -o p -- python $ROOT/"$@" > $OUT/pmc3.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots)
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc4 -o p -- python $ROOT/"$@" > $OUT/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc5 -o p -- python $ROOT/"$@" > $OUT/pmc5.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/pmc_summary.json $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 -name "*counter_collection.csv") > $OUT/pmc_summary.txt 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt | head -20
# keep the merged directory small
find $OUT -name "*.csv" -size +3M -delete
