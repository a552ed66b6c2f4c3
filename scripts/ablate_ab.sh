#!/usr/bin/env bash
# scripts/ablate_ab.sh <config> <patterns> [frames] -- phases of the marching kernel switched off (LANCZOS_DEBUG_SKIP bits:
# 1 no H pass, 2 no fix-up, 4 no V pass, 8 no stores, 16 no loads, 32 no flags, 256 no near flags, 512 no integer flags,
# 1024 V rows copied).  NEEDS the profiling build: make -C lanczos-hls_amd variant VARIANT=prof EXTRA=-DLZ_PROFILE_BITS.
# The library reads its environment once per process (lanczos_env.hpp), so every setting is its own process.
cfg="${1:-c2}"; pats="${2:-gradient}"; frames="${3:-32}"; B=lanczos-hls_amd/build
if ! python3 -c "import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); l.lanczos_version.restype=ctypes.c_char_p; sys.exit(0 if b'profile-bits' in l.lanczos_version() else 1)" "$B/liblanczos_hip_prof.so"; then
  echo "$0: $B/liblanczos_hip_prof.so is missing or is not a -DLZ_PROFILE_BITS build" >&2; exit 2
fi
for s in 0 1 4 5 2 32 256 512; do
  LANCZOS_DEBUG_SKIP=$s python3 scripts/ab.py --config "$cfg" --frames "$frames" --rotate 3 --patterns "$pats" --rounds 3 --steps 20 $B/liblanczos_hip_prof.so 2>&1 | grep "^$cfg" | sed "s/^/skip=$s /"
done
