#!/usr/bin/env bash
mkdir -p gpurun_out/r3j
B=lanczos-hls_amd/build
# ab.py loads every library in ONE process, but the environment is read once per process now: one process per ablation setting
for k in 0 2 512 32; do
  LANCZOS_DEBUG_SKIP=$k python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns noise,blocks,gradient --rounds 3 --steps 20 $B/p_$k.so > gpurun_out/r3j/ab_skip$k.txt 2>&1
  grep "^c2" gpurun_out/r3j/ab_skip$k.txt | sed "s/^/skip=$k /"
done
