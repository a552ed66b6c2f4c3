#!/usr/bin/env bash
mkdir -p gpurun_out/r3l
B=lanczos-hls_amd/build
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient,noise --rounds 5 --steps 20 --mode exact --check $B/idx.so $B/liblanczos_hip_ex7.so > gpurun_out/r3l/ab_exact.txt 2>&1; grep "^c2\|^check" gpurun_out/r3l/ab_exact.txt
