// lfgc_backward.hip -- backward of the fused sample + embed + MLP path (placeholder until the kernels land).
#include "lfgc_common.h"

extern "C" int64_t lfgc_backward_workspace_bytes(const lfgc_mlp_desc* desc, int64_t n_samples) {
    if (!lfgc_mlp_supported(desc)) return LFGC_E_UNSUPPORTED;
    (void)n_samples;
    return 0;
}

extern "C" int lfgc_backward_f32(const lfgc_mlp_desc*, const lfgc_positions*, const float*, int, int, int,
                                 const float*, const float*, const float*, float*, float* const*, float* const*,
                                 float*, void*, int64_t, lfgc_stream_t) {
    return LFGC_E_UNSUPPORTED;
}
