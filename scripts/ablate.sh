#!/usr/bin/env bash
# NEEDS the profiling build: make -C lanczos-hls_amd variant VARIANT=prof EXTRA=-DLZ_PROFILE_BITS and LANCZOS_LIB=.../build/liblanczos_hip_prof.so
# (the production library contains neither the ablation bits nor the switch; the environment is read once per process)
# scripts/ablate.sh "<skip values>" [bench args] -- kernel time with phases of the marching kernel switched off
skips="$1"; shift
# the ablation bits exist only in the profiling build: refuse to print production numbers as an "ablation"
lib="${LANCZOS_LIB:-}"
if [ -z "$lib" ] || ! python3 -c "import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); l.lanczos_version.restype=ctypes.c_char_p; sys.exit(0 if b'profile-bits' in l.lanczos_version() else 1)" "$lib"; then
  echo "$0: set LANCZOS_LIB to a build made with EXTRA=-DLZ_PROFILE_BITS (make -C lanczos-hls_amd variant VARIANT=prof EXTRA=-DLZ_PROFILE_BITS)" >&2; exit 2
fi
for skip in $skips; do
  LANCZOS_DEBUG_SKIP=$skip python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$skip', d['config']['pattern'], 'kernel_us', d['roofline']['kernel_us'])"
done
