#!/usr/bin/env bash
# oracle/build_ref.sh -- compile the REFERENCE's own software path (TEST INFRASTRUCTURE ONLY).
#
# Recipe of SURVEY.md 8(c): the software model lives in /root/reference/LanczosUpscaler/full_TB.h
# lines 29-96 and uses, besides libm, only
#   * `byte`           (full_TB.h:18, `typedef ap_uint<8> byte;` -- here `unsigned char`, the one
#                       substituted line: same clamp-then-truncate store and the same int->double
#                       promotion in `in[i] * kernel`),
#   * MIN / MAX        (lanczos.h:63-64) and SCALE (lanczos.h:112) -- taken from the file itself,
#   * the size macros of the user-written, git-ignored params.h (IN_WIDTH ... NUM_CHANNELS),
#     which the reference tells its user to write (lanczos.h:9-31) -- passed with -D.
# The lines are read from the reference where it lies; the temporary translation unit lives in a
# mktemp directory that is removed afterwards, and ONLY the resulting .so files are written, into
# oracle/_ref/ (git-ignored).  No reference source is copied into the repository.
# The HLS path (lanczos.cpp/worker.cpp/kernel.cpp) needs Xilinx ap_fixed.h/hls_stream.h/hls_math.h,
# which the image lacks: it is unbuildable here and is NOT attempted.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LANCZOS_REFERENCE_DIR:-/root/reference}/LanczosUpscaler"
if [ ! -f "$REF/full_TB.h" ] || [ ! -f "$REF/lanczos.h" ]; then
    echo "build_ref: reference tree not present ($REF) -- nothing to do" >&2
    exit 0
fi
OUT="$HERE/_ref"
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT

{
    echo '#include <math.h>'
    echo '#include <stdint.h>'
    echo '#include <stdio.h>'
    echo '#include <stdlib.h>'
    echo 'typedef unsigned char byte;            /* stands for full_TB.h:18 */'
    sed -n '63,64p;112p' "$REF/lanczos.h"
    sed -n '29,96p' "$REF/full_TB.h"
    cat <<'EOF'
extern "C" void ref_lanczos_expected(const unsigned char* in, unsigned char* out) {
    lanczos_expected((byte (*)[IN_HEIGHT][IN_WIDTH]) in, (byte (*)[OUT_HEIGHT][OUT_WIDTH]) out);
}
extern "C" double ref_lanczos_kernel(double x) { return lanczos_kernel(x); }
extern "C" unsigned char ref_double_to_uint8(double x) { return double_to_uint8(x); }
EOF
} > "$TMP/ref_sw_path.cpp"

n=0
while read -r iw ih ow oh sn sd a c; do
    case "$iw" in ''|\#*) continue ;; esac
    so="$OUT/ref_${iw}x${ih}_${ow}x${oh}_${sn}-${sd}_a${a}_c${c}.so"
    if [ ! -f "$so" ] || [ "$REF/full_TB.h" -nt "$so" ]; then
        g++ -O2 -std=c++14 -w -shared -fPIC \
            -DIN_WIDTH="$iw" -DIN_HEIGHT="$ih" -DOUT_WIDTH="$ow" -DOUT_HEIGHT="$oh" \
            -DSCALE_N="$sn" -DSCALE_D="$sd" -DLANCZOS_A="$a" -DNUM_CHANNELS="$c" \
            "$TMP/ref_sw_path.cpp" -o "$so" -lm
    fi
    n=$((n + 1))
done < "$HERE/ref_configs.txt"
echo "build_ref: $n reference builds in $OUT"
