#!/bin/bash
# register / LDS / spill table of the kernels of one source file whose mangled name contains <filter>:
#   bash tools/kres.sh k_tcn.hip k_tcn_conv_b
cd "$(dirname "$0")/../deepof_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kres.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|LDS Size" | sed -e 's/.*remark: //' -e 's/ \[-Rpass.*//' \
 | awk -v f="$2" '/Function Name/{name=$3; show=index(name,f)>0} show{printf "%s ", $0} /LDS Size/{if(show)print ""}' | sed -e 's/Function Name: _ZN12_GLOBAL__N_1//'
