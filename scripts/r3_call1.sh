#!/usr/bin/env bash
# round 3, call 1: baseline + prefix riding + LDS-DMA placements, then two census dumps of the same binary (run-to-run spread)
set -e
mkdir -p gpurun_out/r3a
B=lanczos-hls_amd/build
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient --rounds 5 --steps 20 --check \
   $B/cur.so $B/ride.so@LANCZOS_RIDE_ALWAYS=1 $B/liblanczos_hip_dma.so $B/liblanczos_hip_dmatop.so > gpurun_out/r3a/ab1.txt 2>&1
cat gpurun_out/r3a/ab1.txt
for i in 1 2 3; do
LANCZOS_STAMP=1 LANCZOS_CENSUS_DUMP=gpurun_out/r3a/census$i.txt python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 --regions 1 > gpurun_out/r3a/census_bench$i.txt 2> gpurun_out/r3a/census_err$i.txt
tail -3 gpurun_out/r3a/census_err$i.txt
done
