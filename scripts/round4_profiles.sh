#!/usr/bin/env bash
# scripts/round4_profiles.sh -- on the GPU box: the round-4 evidence set: bench line + rocprofv3 kernel stats + FETCH/WRITE passes for
# configs 2, 3, 5 (scripts/round_profile.sh), SQ counters for config 2, and the same for LANCZOS_MODE_EXACT on config 2
# (copy into profiles/ with scripts/save_profile.sh round4 / round4_c3 / round4_c5 / round4_exact)
bash scripts/round_profile.sh round4 c2 > gpurun_out/round4_c2.log 2>&1; tail -3 gpurun_out/round4_c2.log | cut -c1-400
bash scripts/round_profile.sh round4_c3 c3 --steps 20 > gpurun_out/round4_c3.log 2>&1; tail -2 gpurun_out/round4_c3.log | cut -c1-300
bash scripts/round_profile.sh round4_c5 c5 --steps 10 > gpurun_out/round4_c5.log 2>&1; tail -2 gpurun_out/round4_c5.log | cut -c1-300
bash scripts/round_profile.sh round4_exact c2 --mode exact --steps 20 > gpurun_out/round4_exact.log 2>&1; tail -2 gpurun_out/round4_exact.log | cut -c1-300
export TMPDIR=/tmp
out=$PWD/gpurun_out/round4_sq; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $out -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --regions 1 --settle-s 0 > $out/bench.json 2> $out/err.txt; echo "sq rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $out/b -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --regions 1 --settle-s 0 > $out/bench2.json 2> $out/err2.txt; echo "sq2 rc=$?"
python3 - $out <<'PY'
import csv,glob,collections,sys,json
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz::k_march" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d={c: round(sum(x)/len(x)) for c,x in agg.items()}
json.dump(d, open(sys.argv[1]+"/sq_summary.json","w"), indent=1)
print(d)
PY
