#!/usr/bin/env bash
mkdir -p gpurun_out/r3f
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "every_integer_scale_instance or oracle_parity_medium" > gpurun_out/r3f/pytest_inst.txt 2>&1
echo "rc=$?"; tail -15 gpurun_out/r3f/pytest_inst.txt
