#!/usr/bin/env bash
# NEEDS the profiling build (-DLZ_PROFILE_BITS); the environment is read once per process: one process per LANCZOS_DEBUG_SKIP value
# scripts/ablate_ab.sh <config> <patterns> -- phases of the marching kernel switched off (LANCZOS_DEBUG_SKIP bits of a
# -DLZ_PROFILE_BITS build), all variants interleaved in ONE process: copies of the library under different names so that
# each gets its own function-local statics.
cfg="$1"; pats="$2"; B=lanczos-hls_amd/build
specs=""
for s in 0 1 4 5 32 256 512; do specs="$specs $B/p_$s.so@LANCZOS_DEBUG_SKIP=$s"; done
python3 scripts/ab.py --config "$cfg" --patterns "$pats" --rounds 5 $specs
