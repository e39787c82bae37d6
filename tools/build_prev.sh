#!/bin/bash
# Same-box A/B helper: build libskp_hip.so with ONE source file taken from another git revision.
#   tools/build_prev.sh <rev> <file under stablekeypoints_amd/csrc> [out.so]   -> tools/csrc/libskp_prev.so
# (tools/ab_build.py then times both builds alternately on the GPU box; built .so files travel with gpurun)
set -e
REV=$1; F=$2; OUT=${3:-tools/csrc/libskp_prev.so}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/stablekeypoints_amd/csrc
make -C $C -j8 > /dev/null
git -C $ROOT show $REV:stablekeypoints_amd/csrc/$F > $C/_prev_$F
(cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -c _prev_$F -o _prev.o \
 && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/$OUT $(ls *.o | grep -v "^${F%.hip}.o$" | grep -v "^_prev.o$") _prev.o)
rm -f $C/_prev_$F $C/_prev.o
ls -la $ROOT/$OUT
