#!/usr/bin/env bash
# scripts/pmc.sh <tag> <lib.so> "<counters>" [bench args] -- one rocprofv3 --pmc pass (GPU box) over bench.py's headline
# launches with the given build of the library; prints the per-launch mean of every counter for the lz:: kernels.
set -uo pipefail
tag="$1"; lib="$2"; ctrs="$3"; shift 3
export TMPDIR=/tmp
out="$PWD/gpurun_out/pmc_$tag"; mkdir -p "$out"
LANCZOS_LIB="$PWD/$lib" timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --regions 1 --settle-s 0.05 "$@" > "$out/bench.json" 2> "$out/err.txt"
echo "$tag rc=$?"
python3 - "$out" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz::" in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
