#!/usr/bin/env bash
mkdir -p gpurun_out/r3h
timeout -k 10 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3h/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3h/pytest_gpu.txt
python3 bench.py > gpurun_out/r3h/bench_default.json 2> gpurun_out/r3h/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3h/bench_default.json')); print(d['ms_per_step'], d['roofline']['frac'], d['other_patterns'], d['host_path_pinned'])"
LANCZOS_LIB=$PWD/lanczos-hls_amd/build/liblanczos_hip_prof.so LANCZOS_STAMP=1 python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 --regions 1 > gpurun_out/r3h/prof.json 2> gpurun_out/r3h/prof.err; echo "prof rc=$?"; grep "STAMP waves\|CENSUS" gpurun_out/r3h/prof.err
