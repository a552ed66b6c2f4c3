#!/usr/bin/env bash
# scripts/round_profile.sh <tag> [config] [extra bench args...] -- on the GPU box: one round's evidence for one
# workload, written under gpurun_out/<tag>/ (copy what should be judged into profiles/):
#   bench.json            python3 bench.py --config <config> ...        -- the line the driver also produces
#   stats/                rocprofv3 --kernel-trace --stats of the same command (headline launches only)
#   pmc_fetch/ pmc_write/ separate --pmc passes (FETCH_SIZE ; WRITE_SIZE) for the HBM traffic figure
#   summary.json          per-kernel average durations + traffic per step (+ the csrc fingerprint it belongs to)
#   traffic_entry.json    the entry bench.py looks up in profiles/traffic.json (key config:frames:pattern:mode)
# rocprofv3 gets the program itself after `--` (python3 bench.py), never a wrapper.
set -uo pipefail
tag="${1:-round}"; cfg="${2:-c2}"; shift; shift || true
extra=("$@")
out="$PWD/gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python3 bench.py --config "$cfg" "${extra[@]}" > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o t -- python3 bench.py --config "$cfg" --no-cpu-baseline --no-extras "${extra[@]}" > "$out/bench_under_rocprof.json" 2> "$out/stats.err"; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python3 bench.py --config "$cfg" --no-cpu-baseline --no-extras --steps 5 --warmup 1 --regions 1 --settle-s 0 "${extra[@]}" > /dev/null 2> "$out/pmc_fetch.err"; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python3 bench.py --config "$cfg" --no-cpu-baseline --no-extras --steps 5 --warmup 1 --regions 1 --settle-s 0 "${extra[@]}" > /dev/null 2> "$out/pmc_write.err"; echo "write rc=$?"
python3 - "$out" "$cfg" <<'PY'
import csv, glob, json, sys, collections, os
out, cfg = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
summ = {"config": cfg, "csrc_fingerprint": bench.csrc_fingerprint(), "kernels": {}, "pmc": {}}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz::" in r["Name"]:
            summ["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                      "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                                      "pct": float(r["Percentage"])}
# traffic per STEP = every lz:: kernel of the step (marching kernel + prefix kernel): sum of per-kernel means
for name in ("pmc_fetch", "pmc_write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "lz::" in r["Kernel_Name"]:
                agg[r["Counter_Name"]][r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for cname, per_kernel in agg.items():
        summ["pmc"][cname] = {k: {"mean_per_launch_KB": sum(v) / len(v), "launches": len(v)} for k, v in per_kernel.items()}
def per_step(counter):
    d = summ["pmc"].get(counter)
    return sum(v["mean_per_launch_KB"] for v in d.values()) if d else None
f, w = per_step("FETCH_SIZE"), per_step("WRITE_SIZE")
try:
    line = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
except Exception:
    line = None
if f and w:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE reports exactly half the bytes of a wide coalesced read on gfx950
    summ["traffic_bytes_per_step"] = {"read_corrected": 2 * f * 1024, "write": w * 1024, "total": (2 * f + w) * 1024,
                                      "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count); separate --pmc passes; "
                                              "sum over the lz:: kernels of one step"}
    if line:
        c = line["config"]
        key = f'{cfg}:{c["frames_per_step_per_gpu"]}:{c["pattern"]}:{c["parity_mode"]}'
        json.dump({key: {"total_bytes_per_step": int((2 * f + w) * 1024), "read_corrected": int(2 * f * 1024),
                         "write": int(w * 1024), "csrc_fingerprint": summ["csrc_fingerprint"],
                         "source": f"scripts/round_profile.sh {os.path.basename(out)} {cfg} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}},
                  open(out + "/traffic_entry.json", "w"), indent=1)
if line:
    summ["bench_roofline"] = line.get("roofline")
    k = summ["kernels"]
    tot = sum(v["avg_ns"] for n, v in k.items() if "lz::" in n)
    if tot:
        summ["frac_recomputed_from_kernel_stats"] = line["roofline"]["algorithmic_bytes_per_step"] / (tot * 1e-9) / 1e9 / 8000.0
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1)[:2500])
PY
cat "$out/bench.json"
