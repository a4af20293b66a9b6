#!/bin/bash
# Prints one line per kernel: name, VGPRs, AGPRs, scratch, occupancy.  Usage: scripts/resusage.sh file.hip
hipcc -O3 --offload-arch=gfx950 -std=c++17 -c -o /dev/null "$1" -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|    VGPRs:|AGPRs:|ScratchSize|Occupancy" \
 | sed -E 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - \
 | sed -E 's/Function Name: //; s/ +/ /g' | while read -r line; do
   name=$(echo "$line" | cut -f1 | c++filt | sed -E 's/\(anonymous namespace\):://; s/\(.*//; s/void //'); rest=$(echo "$line" | cut -f2-); echo "$name | $rest"; done
