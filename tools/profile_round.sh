#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, its rocprofv3 kernel statistics, and the PMC passes the roofline objects'
# `traffic` / instruction / wait figures come from — for the C4 line (headline, default strategy, float64-exact mode) and for
# the other workloads (c3, c5, ogm, filter_lidar, c2).  Output under gpurun_out/$1/; tools/condense_profile.py turns it into the
# files committed under profiles/$1/.  rocprofv3 is always given python3 itself, never a wrapper; no CPU baseline (fork pool) runs
# under the profiler.
set -o pipefail
R=${1:-r04}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "from lidar_slam_amd._lib import csrc_hash; print(csrc_hash())" > $OUT/csrc_sha256.txt || exit 9   # the sources these counters belong to
python3 bench.py > $OUT/bench_default_n1.json 2> $OUT/bench_default_n1.err || exit 1
echo "[profile_round] bench line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 2
echo "[profile_round] stats done"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc() {   # pmc <tag> <bench args…>: three passes (FETCH_SIZE | WRITE_SIZE | SQ counters), --kernel-trace only
  local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$tag -- python3 bench.py "$@" > /dev/null 2> $OUT/pmc_fetch_$tag.err || return 3
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$tag -- python3 bench.py "$@" > /dev/null 2> $OUT/pmc_write_$tag.err || return 4
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_$tag -- python3 bench.py "$@" > /dev/null 2> $OUT/pmc_sq_$tag.err || return 5
  echo "[profile_round] pmc $tag done"
}
pmc c4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs || exit $?
for w in c3 c5 ogm filter_lidar c2; do
  pmc $w --workload $w --no-cpu-baseline || exit $?
done
find $OUT -name "*agent_info.csv" -delete
ls $OUT | head -60
cut -c1-600 $OUT/bench_default_n1.json
