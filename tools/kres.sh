#!/bin/bash
# compact per-kernel resource table of one .hip file: tools/kres.sh file.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kres.o 2>&1 | \
 awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /SGPRs Spill/{ss=$(NF-1)} /VGPRs Spill/{vs=$(NF-1)} /LDS Size/{cmd="c++filt -p " n; cmd | getline d; close(cmd); printf "%-90s vgpr %3s agpr %3s scratch %4s occ %s sspill %3s vspill %3s\n", substr(d,1,90), v,a,s,o,ss,vs} /error/{print}'
