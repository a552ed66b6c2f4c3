#!/usr/bin/env bash
mkdir -p gpurun_out/r3k
B=lanczos-hls_amd/build
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3k/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3k/pytest.txt
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient,noise,blocks --rounds 5 --steps 20 --check $B/cur.so $B/idx.so > gpurun_out/r3k/ab_c2.txt 2>&1; grep "^c2\|^check" gpurun_out/r3k/ab_c2.txt
python3 scripts/ab.py --config c3 --frames 32 --rotate 6 --patterns gradient --rounds 3 --steps 20 --check $B/cur.so $B/idx.so > gpurun_out/r3k/ab_c3.txt 2>&1; grep "^c3\|^check" gpurun_out/r3k/ab_c3.txt
python3 scripts/ab.py --config c5 --frames 8 --rotate 2 --patterns gradient --rounds 3 --steps 10 --check $B/cur.so $B/idx.so > gpurun_out/r3k/ab_c5.txt 2>&1; grep "^c5\|^check" gpurun_out/r3k/ab_c5.txt
python3 scripts/ab.py --config c2 --frames 32 --rotate 3 --patterns gradient --rounds 3 --steps 20 --mode exact --check $B/cur.so $B/idx.so > gpurun_out/r3k/ab_c2_exact.txt 2>&1; grep "^c2\|^check" gpurun_out/r3k/ab_c2_exact.txt
