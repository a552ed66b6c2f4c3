#!/usr/bin/env bash
mkdir -p gpurun_out/r3n
B=lanczos-hls_amd/build
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient,noise --rounds 5 --steps 20 --check $B/base.so $B/liblanczos_hip_late.so > gpurun_out/r3n/ab.txt 2>&1; grep "^c2\|^check" gpurun_out/r3n/ab.txt
