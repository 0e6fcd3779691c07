#!/bin/bash
# usage: tools/kernel_resources.sh file.hip  -> one line per kernel: name vgpr agpr scratch lds occupancy
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|LDS Size|Occupancy|SGPRs:" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n; n=$3; next}{n=n" | "$0}END{print n}' | c++filt 2>/dev/null | cut -c1-260
