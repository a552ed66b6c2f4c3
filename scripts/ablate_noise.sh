#!/usr/bin/env bash
# scripts/ablate_noise.sh [config] [mode] [frames] -- where the noise input's extra time goes: LANCZOS_DEBUG_SKIP bits on the
# profiling build (see scripts/ablate_ab.sh), noise and gradient side by side, one process per setting.
cfg="${1:-c2}"; mode="${2:-lsb1}"; frames="${3:-32}"; B=lanczos-hls_amd/build
if ! python3 -c "import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); l.lanczos_version.restype=ctypes.c_char_p; sys.exit(0 if b'profile-bits' in l.lanczos_version() else 1)" "$B/liblanczos_hip_prof.so"; then
  echo "$0: $B/liblanczos_hip_prof.so is missing or is not a -DLZ_PROFILE_BITS build" >&2; exit 2
fi
for s in 0 2 512 256 32 1 4; do
  LANCZOS_DEBUG_SKIP=$s python3 scripts/ab.py --config "$cfg" --mode "$mode" --frames "$frames" --rotate 3 --patterns noise,gradient --rounds 3 --steps 20 $B/liblanczos_hip_prof.so 2>&1 | grep "^$cfg" | sed "s/^/skip=$s /"
done
