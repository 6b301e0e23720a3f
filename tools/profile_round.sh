#!/bin/bash
# Profiles of the default bench workload for profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>          e.g.  gpurun -- 'tools/profile_round.sh r01_final'
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (same command the bench numbers come from)
# 2. counter passes (separate runs, counters only): FETCH_SIZE ; WRITE_SIZE + TCC hit/miss
# Everything lands in gpurun_out/<tag>/ ; tools/collect_profiles.py copies the summaries to profiles/.
set -e
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--no-cpu-baseline" # the default workload, steps and warm-up of `python bench.py` (100 timed frames after 64); the CPU leg does not touch the GPU
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- python3 $root/bench.py $args > $out/bench_under_rocprof.json 2> $out/trace.log
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o run --output-format csv -- python3 $root/bench.py $args > /dev/null 2> $out/pmc_fetch.log
echo "FETCH_SIZE pass done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc_write -o run --output-format csv -- python3 $root/bench.py $args > /dev/null 2> $out/pmc_write.log
echo "WRITE_SIZE pass done"
# 3. unit-busy passes (what bounds the kernels when it is not HBM): texture address / data units, vector ALU issue.
#    Shorter run (the counters are per launch; 20 timed frames after the 64 learning frames are enough).
short="--no-cpu-baseline --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $out/unit_ta -o run --output-format csv -- python3 $root/bench.py $short > /dev/null 2> $out/unit_ta.log
echo "TA pass done"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum -d $out/unit_td -o run --output-format csv -- python3 $root/bench.py $short > /dev/null 2> $out/unit_td.log
echo "TD pass done"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $out/unit_sq -o run --output-format csv -- python3 $root/bench.py $short > /dev/null 2> $out/unit_sq.log
echo "SQ pass done"
timeout -k 10 300 python3 $root/bench.py > $out/bench.json 2> $out/bench.log
echo "plain bench done"
