#!/bin/bash
# prints VGPR/SGPR/scratch/occupancy/LDS per kernel of a .hip file, one line per kernel
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$1" 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|TotalSGPRs" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: //; s/ \[bytes\/[a-z]+\]//; s/ \[waves\/SIMD\]//' | paste - - - - - - | sed -E 's/\t/  /g'
