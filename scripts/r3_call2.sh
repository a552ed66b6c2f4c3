#!/usr/bin/env bash
# round 3, call 2: self-balancing table vs the static one
set -e
mkdir -p gpurun_out/r3b
B=lanczos-hls_amd/build
LANCZOS_VERBOSE=1 python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient,blocks --rounds 9 --steps 20 --check \
   $B/cur.so $B/bal.so $B/bal0.so@LANCZOS_BALANCE=0 > gpurun_out/r3b/ab.txt 2>&1
grep -v "^lanczos: k_march<" gpurun_out/r3b/ab.txt | tail -40
python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3b/pytest.txt 2>&1 || true
tail -5 gpurun_out/r3b/pytest.txt
