#!/usr/bin/env bash
# scripts/save_profile.sh <tag> -- copy what a round_profile.sh run left under gpurun_out/<tag>/ into profiles/ (tracked):
#   profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (the lz:: kernels' rows + header)
#   profiles/<tag>_rocprof_summary.json  per-kernel averages, PMC traffic, the bench line's roofline object
#   profiles/<tag>_bench.json            the bench.py line of the same run
# and merge the run's traffic entry into profiles/traffic.json.
set -euo pipefail
tag="$1"; src="gpurun_out/$tag"
{ head -1 "$src/stats/t_kernel_stats.csv"; grep 'lz::' "$src/stats/t_kernel_stats.csv" || true; } > "profiles/${tag}_kernel_stats.csv"
cp "$src/summary.json" "profiles/${tag}_rocprof_summary.json"
cp "$src/bench.json" "profiles/${tag}_bench.json"
if [ -f "$src/traffic_entry.json" ]; then
python3 - "$src/traffic_entry.json" <<'PY'
import json, sys
t = json.load(open("profiles/traffic.json"))
t.update(json.load(open(sys.argv[1])))
json.dump(t, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
PY
fi
echo "saved profiles/${tag}_*"
