#!/bin/bash
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
: > $out
for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}
    IFS=',' read -ra envs <<< "$rest"
    line=$(env "${envs[@]}" timeout -k 10 300 python3 $root/bench.py --config5 --no-cpu-baseline --steps 30 2>> $out.err) || { echo "$label FAILED" >> $out; continue; }
    python3 - "$label" "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[2]); print("%-14s %8.1f Msamples/s  %.4f ms/frame" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
done
cat $out
