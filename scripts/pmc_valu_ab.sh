#!/usr/bin/env bash
# scripts/pmc_valu_ab.sh lib1.so lib2.so ... -- VALU occupancy counters of k_march per library (GPU box; rocprofv3 --pmc only)
set -uo pipefail
export TMPDIR=/tmp
for lib in "$@"; do
  out="$PWD/gpurun_out/pmc_valu_$(basename $lib .so)"; mkdir -p "$out"
  for ctrs in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
    LANCZOS_LIB="$PWD/$lib" rocprofv3 --pmc $ctrs --output-format csv -d "$out/$(echo $ctrs | cut -c1-12 | tr ' ' '_')" -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --regions 1 --settle-s 0 > "$out/bench.json" 2> "$out/err.txt"
  done
  python3 - "$out" "$lib" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz::k_march" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {c: round(sum(x)/len(x)) for c,x in sorted(agg.items())})
PY
done
