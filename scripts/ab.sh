#!/usr/bin/env bash
# scripts/ab.sh "<lib1> <lib2> ..." "<patterns>" [rounds] -- interleaved A/B of library builds in ONE process set
# on ONE device (cdna_hip_programming.md 5.4 rule 24): prints kernel_us per build/pattern/round.
libs="$1"; pats="$2"; rounds="${3:-3}"
for r in $(seq 1 "$rounds"); do
  for p in $pats; do
    for l in $libs; do
      LANCZOS_LIB="$PWD/$l" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pattern "$p" 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r', '$l'.split('/')[-1], '$p', 'kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])"
    done
  done
done
