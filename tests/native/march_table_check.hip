#include <hip/hip_runtime.h>
#include <map>
#include <cstdio>
#include "lanczos_hip.h"
#include "lanczos_march.hpp"
using namespace lz;
int check(int strips, int frames, int m_lo, int m_hi, int ms, int taps, int nb, int cus, int nwaves) {
    std::vector<WgEntry> tab; int segs = 0; bool bal = false;
    int n = march_build_table(tab, &segs, strips, frames, m_lo, m_hi, ms, taps, nb, cus, nwaves, &bal);
    std::vector<int> cover((size_t)strips * frames * (m_hi - m_lo), 0);
    int minlen = 1 << 30, maxsegs_used = 0; long minshare = 1 << 30, maxshare = 0; int nonempty = 0;
    for (int b = 0; b < n; b++) {
        long share = 0; int used = 0;
        for (int k = 0; k < segs; k++) {
            WgEntry e = tab[(size_t)b * segs + k];
            if (e.m_b >= e.m_e) continue;
            used++;
            if (e.frame < 0 || e.frame >= frames || e.tx < 0 || e.tx >= strips || e.m_b < m_lo || e.m_e > m_hi) { printf("BAD entry b=%d\n", b); return 1; }
            for (int m = e.m_b; m < e.m_e; m++) cover[((size_t)e.frame * strips + e.tx) * (m_hi - m_lo) + (m - m_lo)]++;
            if (e.m_e - e.m_b < minlen) minlen = e.m_e - e.m_b;
            share += e.m_e - e.m_b;
        }
        if (used) nonempty++;
        if (used > maxsegs_used) maxsegs_used = used;
        if (share && share < minshare) minshare = share;
        if (share > maxshare) maxshare = share;
    }
    for (size_t i = 0; i < cover.size(); i++) if (cover[i] != 1) { printf("COVER %zu = %d (strips %d frames %d)\n", i, cover[i], strips, frames); return 1; }
    printf("strips %3d frames %3d rows [%d,%d) nb %d: n=%d segs=%d(used %d) nonempty=%d balanced=%d share %ld..%ld minlen %d\n", strips, frames, m_lo, m_hi, nb, n, segs, maxsegs_used, nonempty, bal, minshare, maxshare, minlen);
    return 0;
}
int main() {
    int rc = 0;
    for (int f : {1, 2, 3, 5, 8, 16, 17, 24, 32, 33, 48, 64, 68, 100, 128, 200}) rc |= check(15, f, 2, 1080, 12, 6, 4, 256, 6);
    for (int f : {1, 8, 16, 24, 32, 64, 100}) rc |= check(10, f, 1, 720, 12, 6, 2, 256, 6);
    for (int f : {1, 2, 4, 8, 16, 20}) rc |= check(60, f, 3, 2160, 16, 8, 2, 256, 8);
    for (int f : {3, 16}) rc |= check(1, f, 2, 40, 12, 6, 4, 256, 6);
    rc |= check(7, 40, 100, 400, 12, 6, 4, 256, 6);
    printf(rc ? "FAILED\n" : "all partitions exact\n");
    return rc;
}
