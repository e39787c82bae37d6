#!/bin/bash
# register / spill / occupancy report of one kernel source: tools/kres.sh skp_conv_wino4 [grep-pattern]
D=$(dirname "$(readlink -f "$0")")/../stablekeypoints_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -I"$D" -c "$D/$1.hip" -o /tmp/kres_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs:|VGPRs Spill|Occupancy" | paste - - - - - \
  | sed "s/[^ ]*$1.hip:[0-9]*:[0-9]*: *//g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/remark: //g" | grep -E "${2:-.}"
