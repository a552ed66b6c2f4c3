#!/usr/bin/env bash
# scripts/gpu_ab.sh <out-file-under-gpurun_out> <ab.py arguments...> -- runs scripts/ab.py on the GPU box with its output in
# gpurun_out/ (the directory does not travel with the snapshot) and prints the result lines.
out="gpurun_out/$1"; shift
mkdir -p "$(dirname "$out")"
python3 scripts/ab.py "$@" > "$out" 2>&1
rc=$?
grep -E "^c[0-9]|^check|Error|error" "$out" | cut -c1-170
exit $rc
