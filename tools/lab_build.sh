#!/bin/bash
# Lab build of libskp_hip.so with ONE source file compiled with extra -D flags:
#   tools/lab_build.sh <file under stablekeypoints_amd/csrc> <tag> <-D...>   -> build/lab/libskp_<tag>.so
set -e
F=$1; TAG=$2; shift 2
cd "$(dirname "$0")/../stablekeypoints_amd/csrc"
mkdir -p ../../build/lab
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function "$@" -c $F -o ../../build/lab/${F%.hip}_$TAG.o
objs=$(ls *.o | grep -v "^${F%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/lab/libskp_$TAG.so $objs ../../build/lab/${F%.hip}_$TAG.o
ls -la ../../build/lab/libskp_$TAG.so
