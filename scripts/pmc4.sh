#!/usr/bin/env bash
# scripts/pmc4.sh <tag> [bench args]: address-translation and L2 hit/miss PMC passes (GPU box)
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/prof_$tag"; mkdir -p "$out"; export TMPDIR=/tmp
i=0
for ctrs in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_WRITEBACK_sum" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" ; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d "$out/q$i" -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --settle-s 0.05 "$@" > "$out/q$i.json" 2> "$out/q$i.err"
    echo "q$i rc=$?"
done
python3 - "$out" <<'PY'
import csv,glob,collections,sys
for f in sorted(glob.glob(sys.argv[1]+"/q*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "lz::k_march" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(f.split("/")[-2], {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
