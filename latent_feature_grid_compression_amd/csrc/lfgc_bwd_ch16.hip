// backward-kernel instantiations for grid channel stride 16
#include "lfgc_backward.h"
int lfgc_bwd_dispatch_ch16(int MT, const LfgcBwdArgs& a, const LfgcWgradArgs& w, int waves, int h16, int lds_bytes, int grid_data, int grid_w,
                              hipStream_t stream) {
    switch (MT) {
        case 1: return lfgc_launch_bwd<16, 1, 2>(a, w, waves, h16, lds_bytes, grid_data, grid_w, stream);
        case 2: return lfgc_launch_bwd<16, 2, 2>(a, w, waves, h16, lds_bytes, grid_data, grid_w, stream);
        case 4: return lfgc_launch_bwd<16, 4, 2>(a, w, waves, h16, lds_bytes, grid_data, grid_w, stream);
        default: return LFGC_E_UNSUPPORTED;
    }
}
