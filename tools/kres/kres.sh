#!/bin/bash
# usage: tools/kres/kres.sh "NW,R,C,TR,TC,TW,WPE" [extra hipcc flags]   -> VGPRs, spills, occupancy, LDS of that shape
S=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c -Rpass-analysis=kernel-resource-usage \
  -DKRES_SHAPE="$S" "$@" $(dirname $0)/kres.hip -o /tmp/kres.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:|ScratchSize" 
