#!/usr/bin/env bash
mkdir -p gpurun_out/r3i
B=lanczos-hls_amd/build
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "oversized or batch" > gpurun_out/r3i/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3i/pytest.txt
python3 scripts/ab.py --config c2 --frames 64 --rotate 2 --patterns gradient --rounds 3 --steps 10 $B/cur.so lanczos-hls_amd/liblanczos_hip.so > gpurun_out/r3i/ab64.txt 2>&1; grep "^c2" gpurun_out/r3i/ab64.txt
for lib in $B/cur.so lanczos-hls_amd/liblanczos_hip.so; do LANCZOS_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 10 --regions 1 > gpurun_out/r3i/b.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/r3i/b.json')); print('$lib', d['ms_per_step'], d['host_path_pinned'])"; done
