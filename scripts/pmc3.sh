#!/usr/bin/env bash
# scripts/pmc3.sh <tag> [bench args]: store-path / TLB oriented PMC passes (GPU box)
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/prof_$tag"; mkdir -p "$out"; export TMPDIR=/tmp
i=0
for ctrs in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES" \
            "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WRITE_WAVEFRONTS_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
            "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
            "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" ; do
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
