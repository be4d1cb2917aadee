// f16-split forward-kernel instantiations for grid channel stride 32
#include "lfgc_forward16x2.h"
int lfgc_fwd16_dispatch_ch32(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    switch (MT) {
        case 1: return lfgc_launch_fwd16<32, 1, 2>(a, lds_bytes, grid, stream);
        case 2: return lfgc_launch_fwd16<32, 2, 2>(a, lds_bytes, grid, stream);
        case 4: return a.x2 ? lfgc_launch_fwd16x2<32, 4, 2>(a, lds_bytes, grid, stream) : lfgc_launch_fwd16<32, 4, 2>(a, lds_bytes, grid, stream);
        default: return LFGC_E_UNSUPPORTED;
    }
}
