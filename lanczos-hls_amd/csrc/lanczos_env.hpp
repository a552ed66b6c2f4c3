// lanczos_env.hpp -- every environment switch the library knows, read ONCE (first lanczos_create of the process) and listed in
// INTEGRATION.md 7.  Nothing else in csrc/ calls getenv.  Production switches choose between kernels that all produce the same
// results; the profiling switches exist only in builds made with -DLZ_PROFILE_BITS (ablation bits, in-kernel stamps, census).
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

namespace lz {

struct Env {
    bool verbose = false;          // LANCZOS_VERBOSE=1        launch shapes and workgroup tables on stderr
    bool tile_kernel = false;      // LANCZOS_TILE_KERNEL=1    integer scales through the tile-per-workgroup kernel (k_fast) instead of k_march
    bool no_ratp = false;          // LANCZOS_NO_RATP=1        periodic rational scales through the per-index kernel (k_rat)
    bool separate_prefix = false;  // LANCZOS_SEPARATE_PREFIX=1  in-place prefix rows always as their own launch (never riding on k_march)
    int march_wgs = 0;             // LANCZOS_MARCH_WGS=N      marching workgroups per launch (0: one resident round, chosen by the library)
    int march_segs = -1;           // LANCZOS_MARCH_SEGS=0     never the one-workgroup-per-slot table (mode A); >0: always
    std::string rank_weights;      // LANCZOS_RANK_WEIGHTS="w0:w1:w2:w3/v0:v1:v2" slot speeds of the rank-aware shares; "0": equal shares
    bool has_rank_weights = false;
#ifdef LZ_PROFILE_BITS
    int debug_skip = 0;            // LANCZOS_DEBUG_SKIP=bits  ablation bits of the kernels (results are then wrong)
    bool stamp = false;            // LANCZOS_STAMP=1          k_march<u8,3,2,3> with s_memtime stamps + residency census
    std::string census_dump;       // LANCZOS_CENSUS_DUMP=path raw census records of the last stamped launch
#endif
};

inline const Env& env() {
    static const Env e = [] {
        Env v;
        auto flag = [](const char* n) { const char* s = std::getenv(n); return s && std::atoi(s) != 0; };
        auto num = [](const char* n, int dflt) { const char* s = std::getenv(n); return s ? std::atoi(s) : dflt; };
        v.verbose = std::getenv("LANCZOS_VERBOSE") != nullptr;
        v.tile_kernel = flag("LANCZOS_TILE_KERNEL");
        v.no_ratp = flag("LANCZOS_NO_RATP");
        v.separate_prefix = flag("LANCZOS_SEPARATE_PREFIX");
        v.march_wgs = num("LANCZOS_MARCH_WGS", 0);
        v.march_segs = num("LANCZOS_MARCH_SEGS", -1);
        if (const char* s = std::getenv("LANCZOS_RANK_WEIGHTS")) v.rank_weights = s, v.has_rank_weights = true;
#ifdef LZ_PROFILE_BITS
        v.debug_skip = num("LANCZOS_DEBUG_SKIP", 0);
        v.stamp = flag("LANCZOS_STAMP");
        if (const char* s = std::getenv("LANCZOS_CENSUS_DUMP")) v.census_dump = s;
#endif
        return v;
    }();
    return e;
}

}  // namespace lz
