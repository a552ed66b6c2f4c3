#!/usr/bin/env bash
# scripts/round_profile.sh <tag> -- on the GPU box: the round's evidence, written under gpurun_out/<tag>/
#   bench.json            python bench.py (default workload) -- the line the driver will also produce
#   stats/                rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch/ pmc_write/ separate --pmc passes (FETCH_SIZE ; WRITE_SIZE) for the HBM traffic figure
set -uo pipefail
tag="${1:-round}"
out="$PWD/gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o t -- python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.err"; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2> "$out/pmc_fetch.err"; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2> "$out/pmc_write.err"; echo "write rc=$?"
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summ = {"kernels": {}, "pmc": {}}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz::" in r["Name"]:
            summ["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                      "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                                      "pct": float(r["Percentage"])}
for name in ("pmc_fetch", "pmc_write"):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "lz::k_march" in r["Kernel_Name"] or "lz::k_fast" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summ["pmc"][k] = {"mean_per_launch_KB": sum(v) / len(v), "launches": len(v)}
f = summ["pmc"].get("FETCH_SIZE", {}).get("mean_per_launch_KB")
w = summ["pmc"].get("WRITE_SIZE", {}).get("mean_per_launch_KB")
if f and w:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE reports exactly half the bytes of a wide coalesced read on gfx950
    summ["traffic_bytes_per_launch"] = {"read_corrected": 2 * f * 1024, "write": w * 1024, "total": (2 * f + w) * 1024,
                                        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count); separate --pmc passes"}
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1)[:1800])
PY
cat "$out/bench.json"
