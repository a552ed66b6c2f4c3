#!/usr/bin/env bash
set -e
mkdir -p gpurun_out/r3d
B=lanczos-hls_amd/build
python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3d/pytest.txt 2>&1 || (tail -30 gpurun_out/r3d/pytest.txt; exit 1)
tail -3 gpurun_out/r3d/pytest.txt
LANCZOS_VERBOSE=1 python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient --rounds 9 --steps 20 --check \
   $B/cur.so $B/fill.so $B/nofill.so@LANCZOS_PREFIX_FILL=0 $B/fill.so@LANCZOS_BALANCE=0 > gpurun_out/r3d/ab.txt 2>&1
grep -v "^lanczos: k_march<" gpurun_out/r3d/ab.txt | tail -24
python3 scripts/ab.py --config c2 --frames 16 --rotate 6 --patterns gradient --rounds 5 --steps 20 --check \
   $B/cur.so $B/fill.so > gpurun_out/r3d/ab16.txt 2>&1
tail -3 gpurun_out/r3d/ab16.txt
