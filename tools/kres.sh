#!/bin/bash
# Per-kernel resource usage of the device code (VGPRs, spills, scratch, LDS, occupancy) as hipcc reports it.
# usage: tools/kres.sh [extra hipcc flags, e.g. -DMI_FW=7]
cd "$(dirname "$0")/../mve_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c dmrecon_device.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error| Name:|    VGPRs:|ScratchSize|Occupancy|LDS Size|VGPRs Spill" | sed 's/.*remark: //;s/^[^ ]* *//;s/\[-Rpass.*//' \
 | awk '/Name:/{if(l)print l; l=$0; next}{l=l" |"$0}END{print l}' | sed 's/  */ /g'
