#!/usr/bin/env bash
set -e
mkdir -p gpurun_out/r3e
for pat in gradient noise; do
LANCZOS_STAMP=1 python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 --regions 1 --pattern $pat > gpurun_out/r3e/stamp_$pat.json 2> gpurun_out/r3e/stamp_$pat.txt
grep "STAMP\|CENSUS" gpurun_out/r3e/stamp_$pat.txt
done
