// forward-kernel instantiations for grid channel stride 32 (split per stride to compile in parallel)
#include "lfgc_forward.h"
int lfgc_fwd_dispatch_ch32(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    switch (MT) {
        case 1: return lfgc_launch_fwd<32, 1, 2>(a, lds_bytes, grid, stream);
        case 2: return lfgc_launch_fwd<32, 2, 2>(a, lds_bytes, grid, stream);
        case 4: return lfgc_launch_fwd<32, 4, 2>(a, lds_bytes, grid, stream);
        default: return LFGC_E_UNSUPPORTED;
    }
}
