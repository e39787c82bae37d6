#!/bin/bash
# Lab builds of libskp_hip.so with parts of the split convolution's stage loop removed (-DW4S_ABL=<mask>: 1 side jobs,
# 2 filter loads, 4 MFMAs, 8 LDS operand reads; results are WRONG, only the time is of interest) -> build/abl/libskp_abl<mask>.so
set -e
cd "$(dirname "$0")/../stablekeypoints_amd/csrc"
mkdir -p ../../build/abl
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -DW4S_ABL=$m -c skp_conv_wino4s.hip -o ../../build/abl/w4s_$m.o
  objs=$(ls *.o | grep -v skp_conv_wino4s.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/abl/libskp_abl$m.so $objs ../../build/abl/w4s_$m.o
done
