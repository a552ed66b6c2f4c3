#!/usr/bin/env bash
set -e
mkdir -p gpurun_out/r3c
B=lanczos-hls_amd/build
LANCZOS_BALANCE_DUMP=gpurun_out/r3c/bal LANCZOS_VERBOSE=1 python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient --rounds 9 --steps 20 \
   $B/bal.so > gpurun_out/r3c/ab.txt 2>&1
grep -v "^lanczos: k_march<" gpurun_out/r3c/ab.txt | tail -12
