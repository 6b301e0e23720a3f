#!/bin/bash
# One rocprofv3 counter pass over a short bench run (counters only; no tracing domains beside them).
#   tools/pmc_pass.sh <out-name> <counter> [<counter> ...]
# Writes gpurun_out/<out-name>/ ; summarise with tools/pmc_summary.py.
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
exec rocprofv3 --pmc "$@" -d /root/repo/gpurun_out/$name -o run --output-format csv -- python3 /root/repo/bench.py --steps 4 --warmup 3 --no-cpu-baseline
