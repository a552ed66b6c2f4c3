#!/usr/bin/env bash
# scripts/profile_gpu.sh <tag> [bench args...] -- run on the GPU box (via gpurun).
# 1. rocprofv3 --kernel-trace --stats of the bench command  -> gpurun_out/prof_<tag>/stats
# 2. separate --pmc passes (never combined with traces)      -> gpurun_out/prof_<tag>/pmc*
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/prof_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o t -- python3 bench.py --no-cpu-baseline "${args[@]}" > "$out/bench_stats.json" 2> "$out/stats.err"
echo "stats rc=$?"
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc$i" -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "${args[@]}" > "$out/bench_pmc$i.json" 2> "$out/pmc$i.err"
    echo "pmc$i rc=$?"
done
find "$out" -name "*.csv" | head -40
