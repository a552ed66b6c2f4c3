#!/usr/bin/env bash
# scripts/sweep_env.sh <settings file> [ab.py args...] -- one process per line of the settings file (the library reads its
# environment once per process): each line is a list of NAME=VALUE assignments (or "-" for the defaults); the whole-step time
# of every setting is printed beside it.  Runs on the GPU box (gpurun); a baseline line ("-") every few settings shows the drift.
set -uo pipefail
file="$1"; shift
while IFS= read -r line; do
  case "$line" in ''|\#*) continue ;; esac
  if [ "$line" = "-" ]; then envs=(); else read -r -a envs <<< "$line"; fi
  out=$(env "${envs[@]}" timeout -k 10 120 python3 scripts/ab.py "$@" lanczos-hls_amd/liblanczos_hip.so 2>/dev/null | grep "median")
  echo "$line | $out"
done < "$file"
