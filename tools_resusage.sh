#!/bin/bash
# prints VGPR/SGPR/scratch/occupancy per kernel of a .hip file
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$1" 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|TotalSGPRs" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - 
