#!/usr/bin/env bash
mkdir -p gpurun_out/r3g
B=lanczos-hls_amd/build
timeout -k 10 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3g/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3g/pytest_gpu.txt
cp lanczos-hls_amd/liblanczos_hip.so $B/new.so
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient,noise,blocks --rounds 5 --steps 20 --check $B/cur.so $B/new.so > gpurun_out/r3g/ab_c2.txt 2>&1; grep "^c2\|^check" gpurun_out/r3g/ab_c2.txt
python3 scripts/ab.py --config c3 --frames 32 --rotate 6 --patterns gradient --rounds 3 --steps 20 --check $B/cur.so $B/new.so > gpurun_out/r3g/ab_c3.txt 2>&1; grep "^c3\|^check" gpurun_out/r3g/ab_c3.txt
python3 scripts/ab.py --config c5 --frames 8 --rotate 2 --patterns gradient --rounds 3 --steps 10 --check $B/cur.so $B/new.so > gpurun_out/r3g/ab_c5.txt 2>&1; grep "^c5\|^check" gpurun_out/r3g/ab_c5.txt
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient --rounds 3 --steps 20 --mode exact --check $B/cur.so $B/new.so > gpurun_out/r3g/ab_c2_exact.txt 2>&1; grep "^c2\|^check" gpurun_out/r3g/ab_c2_exact.txt
python3 scripts/instance_speed.py > gpurun_out/r3g/instance_speed.txt 2>&1; cat gpurun_out/r3g/instance_speed.txt
python3 bench.py > gpurun_out/r3g/bench_default.json 2> gpurun_out/r3g/bench_default.err; tail -c 3000 gpurun_out/r3g/bench_default.json
