#!/bin/bash
# VGPRs / scratch / spills per kernel of one .hip file (compile only, no GPU needed): tools/kernel_resources.sh lslam_icp_batch.hip
cd "$(dirname "$0")/../lidar_slam_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 |
  awk '/Function Name/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)} /VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /VGPRs Spill/ {print n, "vgpr", v, "scratch", s, "occ", o, "spill", $(NF-1)}' |
  while read n rest; do echo "$(echo $n | c++filt | sed 's/lslam:://; s/(.*//; s/void //; s/, /,/g') $rest"; done | sort
