#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, its rocprofv3 kernel statistics, and the PMC passes the
# roofline object's `traffic` / instruction counts come from.  Output under gpurun_out/$1/; tools/condense_profile.py
# turns it into the files committed under profiles/$1/.
set -o pipefail
R=${1:-r02}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "from lidar_slam_amd._lib import csrc_hash; print(csrc_hash())" > $OUT/csrc_sha256.txt || exit 9   # the sources these counters belong to
python3 bench.py > $OUT/bench_default_n1.json 2> $OUT/bench_default_n1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 2
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err || exit 4
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2> $OUT/pmc_sq.err || exit 5
find $OUT -name "*agent_info.csv" -delete
ls -R $OUT | head -40
cat $OUT/bench_default_n1.json | cut -c1-600
