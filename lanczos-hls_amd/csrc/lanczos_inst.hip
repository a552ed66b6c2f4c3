// lanczos_inst.hip -- one group of instantiated configurations of the integer-scale kernels (k_march, k_fast).
// Compiled once per group (-DLZ_INST_GROUP=0..3, see lanczos_fast.hpp: LZ_FAST_CONFIGS_G*) so that the 35 configurations of
// the reference's params.h space (lanczos.h:9-31) build in parallel translation units instead of one ten-minute one.
#include <hip/hip_runtime.h>

#include "../../include/lanczos_hip.h"
#include "lanczos_march.hpp"

#ifndef LZ_INST_GROUP
#error "compile with -DLZ_INST_GROUP=0..3"
#endif
#define LZ_CAT2(a, b) a##b
#define LZ_CAT(a, b) LZ_CAT2(a, b)
#define LZ_GROUP_CONFIGS LZ_CAT(LZ_FAST_CONFIGS_G, LZ_INST_GROUP)

namespace lz {

hipError_t LZ_CAT(march_launch_g, LZ_INST_GROUP)(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const FastConsts& fc,
                                                 hipStream_t stream, bool* prefix_fused, WgTabCache* cache, bool query_only) {
#define X(T, C, S, A)                                                                               \
    if (d.bytes_per_sample == (int)sizeof(T) && d.channels == C && d.scale_n == S && d.a == A)      \
        return march_launch_t<T, C, S, A>(d, g, t, fc, stream, prefix_fused, query_only, cache);
    LZ_GROUP_CONFIGS(X)
#undef X
    return hipErrorNotSupported;
}

hipError_t LZ_CAT(fast_launch_g, LZ_INST_GROUP)(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const FastConsts& fc,
                                                hipStream_t stream) {
#define X(T, C, S, A)                                                                               \
    if (d.bytes_per_sample == (int)sizeof(T) && d.channels == C && d.scale_n == S && d.a == A)      \
        return fast_launch_t<T, C, S, A>(d, g, t, fc, stream);
    LZ_GROUP_CONFIGS(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace lz
