#!/usr/bin/env bash
# NEEDS the profiling build for the skip=8 leg (LANCZOS_LIB=.../build/liblanczos_hip_prof.so)
# scripts/pmc5.sh: VALU occupancy counters for the full kernel and the no-store ablation (GPU box)
set -uo pipefail
export TMPDIR=/tmp
# the ablation bits exist only in the profiling build: refuse to print production numbers as an "ablation"
lib="${LANCZOS_LIB:-}"
if [ -z "$lib" ] || ! python3 -c "import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); l.lanczos_version.restype=ctypes.c_char_p; sys.exit(0 if b'profile-bits' in l.lanczos_version() else 1)" "$lib"; then
  echo "$0: set LANCZOS_LIB to a build made with EXTRA=-DLZ_PROFILE_BITS (make -C lanczos-hls_amd variant VARIANT=prof EXTRA=-DLZ_PROFILE_BITS)" >&2; exit 2
fi
for sk in 0 8; do
  out="$PWD/gpurun_out/prof_valu_$sk"; mkdir -p "$out"
  LANCZOS_DEBUG_SKIP=$sk timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d "$out" -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --settle-s 0.05 --pattern blocks > "$out/bench.json" 2> "$out/err.txt"
  echo "skip=$sk rc=$?"
  python3 - "$out" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz::k_march" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(x)/len(x)) for c,x in agg.items()})
PY
done
