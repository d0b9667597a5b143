#!/bin/bash
# register / scratch use of every kernel of one source file: bash tools/kernel_regs.sh dcn_forward_plane [extra flags]
f=$1; shift
cd "$(dirname "$0")/../kgdet_amd/csrc" || exit 1
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kregs_$f.o 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /LDS Size/ {printf "%-90s vgpr %s agpr %s scratch %s\n", name, v, a, s}' | c++filt | cut -c1-170
