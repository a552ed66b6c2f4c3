#!/usr/bin/env bash
# scripts/pmc2.sh <tag> [bench args]: stall-oriented PMC passes (GPU box)
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/prof_$tag"; mkdir -p "$out"; export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" \
            "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_CYCLES" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" ; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d "$out/q$i" -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$out/q$i.json" 2> "$out/q$i.err"
    echo "q$i rc=$?"
done
python3 - "$out" <<'PY'
import csv,glob,collections,sys
for f in sorted(glob.glob(sys.argv[1]+"/q*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "lz::k_march" in r["Kernel_Name"] or "lz::k_fast" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(f.split("/")[-2], {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
