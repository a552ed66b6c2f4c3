#!/usr/bin/env bash
mkdir -p gpurun_out/r3q
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3q/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3q/pytest_gpu.txt
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 --regions 2 --no-cpu-baseline > gpurun_out/r3q/rehearsal_2ranks.json 2> gpurun_out/r3q/rehearsal.err; echo "rehearsal rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r3q/rehearsal_2ranks.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], list(d.keys())[-6:])"
