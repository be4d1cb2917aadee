// f16-split forward-kernel instantiations for grid channel stride 16
#include "lfgc_forward16.h"
int lfgc_fwd16_dispatch_ch16(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    switch (MT) {
        case 1: return lfgc_launch_fwd16<16, 1, 2>(a, lds_bytes, grid, stream);
        case 2: return lfgc_launch_fwd16<16, 2, 2>(a, lds_bytes, grid, stream);
        case 4: return lfgc_launch_fwd16<16, 4, 2>(a, lds_bytes, grid, stream);
        default: return LFGC_E_UNSUPPORTED;
    }
}
