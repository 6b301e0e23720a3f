#!/bin/bash
# A/B of library builds and / or properties on the default bench workload, within ONE gpurun call (boxes of the pool differ
# by +-2 %):  tools/ab_bench.sh <out file> <label>=<lib or ->[,<bench args>] ...
#   e.g. tools/ab_bench.sh gpurun_out/ab.txt base=- lc0=lib/libmqhip_lc0.so "lock=-,--prop LC try-lock=1"
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
: > $out
for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}
    lib=${rest%%,*}; args=""
    [ "$rest" != "$lib" ] && args=${rest#*,}
    [ "$lib" = "-" ] && lib=lib/libmqhip.so
    IFS=',' read -ra extra <<< "$args"
    line=$(MQHIP_LIB=$root/merian-quake_amd/$lib timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --steps 60 "${extra[@]}" 2>> $out.err) || { echo "$label FAILED" >> $out; continue; }
    python3 - "$label" "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-14s %8.1f Msamples/s  %.4f ms/frame  kernels %s  launches %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernels_ms_per_frame"], r["launches_ms"]))
PY
done
cat $out
