set -u
for r in 1 2 3; do for sk in 0 4096 8192; do LANCZOS_DEBUG_SKIP=$sk python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r skip=$sk', 'kernel_us', d['roofline']['kernel_us'], {k:v['kernel_us'] for k,v in d['other_patterns'].items()})"; done; done
