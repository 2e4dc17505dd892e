#!/bin/bash
# Phase shares and search statistics of the task-queue kernel (profile build: make LIBDIR=../lib_prof EXTRA=-DLSLAM_QUEUE_PROFILE).
# usage on the GPU box: tools/queue_stats.sh [cfg ...]   → gpurun_out/queue_stats_{c4,c3}.log
export LSLAM_LIB=$PWD/lidar_slam_amd/lib_prof/liblslam_hip.so LSLAM_DEBUG=1
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/prune_ab.py "${@:-default}" 2>&1 | grep -v "previous launch" | awk '!seen[$0]++' > gpurun_out/queue_stats_c4.log
timeout -k 10 200 python3 tools/c3_probe.py "${@:-default}" 2>&1 | grep -v "previous launch" | awk '!seen[$0]++' > gpurun_out/queue_stats_c3.log
C3_F64=1 timeout -k 10 200 python3 tools/c3_probe.py "${@:-default}" 2>&1 | grep -v "previous launch" | awk '!seen[$0]++' > gpurun_out/queue_stats_c3f64.log
