// lanczos_fast.hpp -- specialised kernels (integer scale, LDS-staged tiles).  Stub for bring-up.
#pragma once
#include "lanczos_kernels_common.hpp"
#include "lanczos_taps.hpp"

namespace lz {
struct FastConsts { int dummy; };
inline bool fast_prepare(const lanczos_desc&, const AxisTaps&, const AxisTaps&, FastConsts*) { return false; }
inline bool fast_supports(const lanczos_desc&, const FrameGeom&) { return false; }
inline hipError_t fast_launch(const lanczos_desc&, const FrameGeom&, const TapTables&, const FastConsts&,
                              hipStream_t) { return hipErrorNotSupported; }
}  // namespace lz
