#!/bin/bash
# tools/ab_env.sh <out> <label>=<ENV=VAL[,ENV=VAL]> ...  : default bench workload under different environment settings, one gpurun call
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
: > $out
for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}
    IFS=',' read -ra envs <<< "$rest"
    line=$(env "${envs[@]}" timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --steps 60 2>> $out.err) || { echo "$label FAILED" >> $out; continue; }
    python3 - "$label" "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-14s %8.1f Msamples/s  %.4f ms/frame  kernels %s  launches %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernels_ms_per_frame"], r["launches_ms"]))
PY
done
cat $out
