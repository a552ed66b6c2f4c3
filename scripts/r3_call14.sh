#!/usr/bin/env bash
mkdir -p gpurun_out/r3m
timeout -k 10 420 python3 tests/soak_gpu.py 330 0 > gpurun_out/r3m/soak_u8.txt 2>&1; echo "u8 rc=$?"; tail -2 gpurun_out/r3m/soak_u8.txt
timeout -k 10 300 python3 tests/soak_gpu.py 200 1.0 > gpurun_out/r3m/soak_u16.txt 2>&1; echo "u16 rc=$?"; tail -2 gpurun_out/r3m/soak_u16.txt
