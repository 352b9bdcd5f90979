#!/bin/bash
# Per-kernel resource usage of the device code (VGPRs, spills, scratch, LDS, occupancy) as hipcc reports it.
cd "$(dirname "$0")/../mve_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c dmrecon_device.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error|Function Name|    VGPRs:|ScratchSize|Occupancy|LDS Size|VGPRs Spill" | sed 's/.*remark: [^ ]* *//;s/\[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" |"$0}END{print l}' | sed 's/Function Name: //;s/  */ /g'
