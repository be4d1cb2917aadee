// lfgc_capi_forward.hip -- C-ABI entry for the fused forward: argument checks, plan, dispatch.
#include <stdlib.h>
#include "lfgc_forward.h"

int lfgc_fwd_dispatch_ch8(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd_dispatch_ch16(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd_dispatch_ch24(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd_dispatch_ch32(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd16_dispatch_ch8(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd16_dispatch_ch16(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd16_dispatch_ch24(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);
int lfgc_fwd16_dispatch_ch32(int MT, const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream);

namespace {
int num_cus() { return lfgc_num_cus(); }
#ifdef LFGC_STAMPS
unsigned long long* g_stamps = nullptr;
#endif
}  // namespace

__global__ void lfgc_clear_word_kernel(int32_t* w) {
    if (threadIdx.x == 0) __hip_atomic_store(w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef LFGC_STAMPS
// Diagnostics builds only (tools/phase_stamps.py): device buffer of 16 counters per wave slot (grid x 8 waves).
extern "C" void lfgc_debug_set_stamp_buffer(void* p) { g_stamps = reinterpret_cast<unsigned long long*>(p); }
#endif

// Validates `positions` and fills the position part of the kernel arguments; returns the sample count
// through *n_out.
int lfgc_fill_positions(const lfgc_positions* ps, LfgcFwdArgs* a, long long* n_out) {
    if (!ps) return LFGC_E_NULL;
    if (ps->pos) {
        if (ps->n < 0) return LFGC_E_SHAPE;
        a->pos = ps->pos;
        *n_out = ps->n;
        a->res0 = a->res1 = a->res2 = 2; a->x_begin = 0; a->tile = 32;
        a->scale0 = a->scale1 = a->scale2 = 1.0f;
        return LFGC_OK;
    }
    if (ps->res[0] < 2 || ps->res[1] < 2 || ps->res[2] < 2 || ps->tile < 1) return LFGC_E_SHAPE;
    if (ps->x_begin < 0 || ps->x_end > ps->res[0] || ps->x_end < ps->x_begin) return LFGC_E_SHAPE;
    a->pos = nullptr;
    a->res0 = ps->res[0]; a->res1 = ps->res[1]; a->res2 = ps->res[2];
    a->x_begin = ps->x_begin; a->tile = ps->tile;
    // dataset.scales = max_idx / max(max_idx)     (data/IndexDataset.py:64-65), fp32 division
    const float m0 = (float)(ps->res[0] - 1), m1 = (float)(ps->res[1] - 1), m2 = (float)(ps->res[2] - 1);
    const float mm = m0 > m1 ? (m0 > m2 ? m0 : m2) : (m1 > m2 ? m1 : m2);
    a->scale0 = m0 / mm; a->scale1 = m1 / mm; a->scale2 = m2 / mm;
    *n_out = (long long)(ps->x_end - ps->x_begin) * ps->res[1] * ps->res[2];
    return LFGC_OK;
}

extern "C" int lfgc_forward_f32(const lfgc_mlp_desc* desc, const lfgc_positions* positions,
                                const float* grid_cl, int D, int H, int W,
                                const float* packed, int precision, int clamp, float* out, float* stash,
                                int32_t* status, lfgc_stream_t stream) {
    if (!desc || !positions || !grid_cl || !packed || !out) return LFGC_E_NULL;
    if (!lfgc_mlp_supported(desc)) return LFGC_E_UNSUPPORTED;
    if (D < 1 || H < 1 || W < 1) return LFGC_E_SHAPE;
    if ((long long)D * H * W * lfgc_roundup(desc->grid_channels, 8) >= (1LL << 30)) return LFGC_E_UNSUPPORTED;   // 32-bit byte offsets
    if (precision != LFGC_PRECISION_F32 && precision != LFGC_PRECISION_F16X2 && precision != LFGC_PRECISION_F16) return LFGC_E_UNSUPPORTED;
    if ((((uintptr_t)grid_cl) | ((uintptr_t)packed)) & 15) return LFGC_E_ALIGN;
    const LfgcPlan p = lfgc_make_plan(desc->grid_channels, desc->hidden, desc->num_layers, desc->n_freqs);
    LfgcFwdArgs a;
    long long n = 0;
    const int rc = lfgc_fill_positions(positions, &a, &n);
    if (rc != LFGC_OK) return rc;
    if (n == 0) return LFGC_OK;
    a.n = n;
    a.grid = grid_cl; a.D = D; a.H = H; a.W = W; a.Cs = p.CH;
    a.packed = packed; a.L = p.L; a.clamp = clamp; a.out = out; a.stash = stash;
    // LDS: [Wf | bf] + every layer block (resident: 4-wave workgroups, two per CU) or a 2-deep ring of the
    // largest block (streamed: 8-wave workgroups, one per CU).  The stash is laid out per 32-sample tile in
    // whole 128-sample groups either way (lfgc_stash_bytes), so both builds write the same format.
    const bool h16 = precision != LFGC_PRECISION_F32;
    a.single = precision == LFGC_PRECISION_F16 ? 1 : 0;
    a.status = nullptr; a.redo_if = nullptr; a.stamps = nullptr;
#ifdef LFGC_STAMPS
    a.stamps = g_stamps;
#endif
    const int all_blocks = h16 ? p.blkh0 + (p.L - 1) * p.blkh1 : p.off_final;
    const int max_block = h16 ? (p.blkh0 > p.blkh1 ? p.blkh0 : p.blkh1) : (p.blk0 > p.blk1 ? p.blk0 : p.blk1);
    const int fixed = p.HP + 4 + (h16 ? 16 + LFGC_MAX_LAYERS * p.HP : 0);   // [Wf | bf] (+ per-layer scales + resident biases)
    a.resident = ((fixed + all_blocks) * 4 <= 80 * 1024) ? 1 : 0;
    int lds_bytes = (fixed + (a.resident ? all_blocks : 2 * max_block)) * 4;
    a.coord_table = 0;
    if (!a.pos) {                                       // per-axis coordinate tables behind the weight region
        const long long tbl = 4LL * ((long long)a.res0 + a.res1 + a.res2);
        const long long cap = a.resident ? 80 * 1024 : 160 * 1024;
        if (lds_bytes + tbl <= cap) { a.coord_table = 1; lds_bytes += (int)tbl; }
    }
    // streamed nets: 8-wave workgroups once every CU gets at least one 256-sample batch, else 4-wave ones
    a.waves = (!a.resident && (n + 255) / 256 >= num_cus()) ? 8 : 4;
    if (const char* e = getenv("LFGC_FWD_WAVES")) { if (!a.resident && (e[0] == '4' || e[0] == '8')) a.waves = e[0] - '0'; }   // diagnostics
    // always whole 256-sample groups of tiles, so the stash covers the same tile range whichever build runs
    a.nbatches = (n + 255) / 256 * (8 / a.waves);
    // Lattice mode on the f16 builds: z-run tiles + column sampler (lfgc_forward.h) when the column a 32-voxel run touches
    // is short (volume at least ~3x finer than the grid along z: every BASELINE full-volume shape) and fits the LDS left.
    a.zrun = 0; a.nzc = 2; a.tiles_per_row = 1; a.ntiles = 0; a.x2 = 0;
    if (h16 && !a.pos && a.coord_table && !stash && !getenv("LFGC_NO_ZRUN")) {
        const int nzc = (int)(31.0 * (double)D / (double)(a.res2 - 1) + 1e-3) + 3;
        const long long rows = (long long)(positions->x_end - positions->x_begin) * a.res1;
        const int tpr = (a.res2 + LFGC_TILE_SAMPLES - 1) / LFGC_TILE_SAMPLES;
        const long long ntiles = rows * tpr;
        const long long tbl4 = 4LL * (((long long)a.res0 + a.res1 + a.res2 + 3) & ~3LL) - 4LL * ((long long)a.res0 + a.res1 + a.res2);
        const long long cap = a.resident ? 80 * 1024 : 160 * 1024;
        const long long col = 4LL * 8 * nzc * (p.CH + 4);            // 8 waves x nzc padded rows (LfgcColumnSampler::CS)
        if (nzc <= 12 && ntiles < (1LL << 31) && lds_bytes + tbl4 + col <= cap) {
            a.zrun = 1; a.nzc = nzc; a.tiles_per_row = tpr; a.ntiles = ntiles;
            lds_bytes += (int)(tbl4 + col);
            a.waves = (!a.resident && (ntiles + 7) / 8 >= num_cus()) ? 8 : 4;
            if (const char* e = getenv("LFGC_FWD_WAVES")) { if (!a.resident && (e[0] == '4' || e[0] == '8')) a.waves = e[0] - '0'; }
            a.nbatches = (ntiles + a.waves - 1) / a.waves;
            // experimental (LFGC_FWD_X2=1): two tiles per wave, one wave per SIMD (lfgc_forward16x2.h; 32 channels x 128 wide)
            if (!a.resident && p.CH == 32 && p.MT == 4 && (ntiles + 7) / 8 >= num_cus() && getenv("LFGC_FWD_X2")) {
                a.x2 = 1; a.waves = 4; a.nbatches = (ntiles + 7) / 8;
            }
        }
    }
    long long grid = (a.resident ? 2LL : 1LL) * num_cus();
    if (grid > a.nbatches) grid = a.nbatches;
    hipStream_t st = (hipStream_t)stream;
    if (h16) {
        if (status) {
            // range screen on: cleared here, set by the kernel, read by the predicated redo -- all in stream order.
            // Cleared by a kernel of our own, NOT hipMemsetAsync: inside a captured HIP graph (ROCm 7.2) the memset
            // node of a 4-byte clear did not take effect before the following kernel nodes on replay -- the redo then
            // ran in full on every replay of a captured train step (measured: 58 us instead of 4 us; profiles/r2).
            hipLaunchKernelGGL(lfgc_clear_word_kernel, dim3(1), dim3(64), 0, st, status);
            LFGC_HIP_CHECK_LAUNCH();
            a.status = status;
        }
        int rc16;
        switch (p.CH) {
            case 8: rc16 = lfgc_fwd16_dispatch_ch8(p.MT, a, lds_bytes, (int)grid, st); break;
            case 16: rc16 = lfgc_fwd16_dispatch_ch16(p.MT, a, lds_bytes, (int)grid, st); break;
            case 24: rc16 = lfgc_fwd16_dispatch_ch24(p.MT, a, lds_bytes, (int)grid, st); break;
            case 32: rc16 = lfgc_fwd16_dispatch_ch32(p.MT, a, lds_bytes, (int)grid, st); break;
            default: return LFGC_E_UNSUPPORTED;
        }
        if (rc16 != LFGC_OK || !status) return rc16;
        // Range fallback: the same pass on the exact-fp32 build, enqueued behind the fast one; its workgroups return
        // at once unless the fast kernel has set *status (a sample left the f16 range: diverged or very wide model).
        // No host synchronisation, graph-capturable; costs one empty launch when nothing overflowed.
        a.status = nullptr; a.redo_if = status; a.single = 0; a.zrun = 0; a.x2 = 0;
        const int all32 = p.off_final, max32 = p.blk0 > p.blk1 ? p.blk0 : p.blk1, fixed32 = p.HP + 4;
        a.resident = ((fixed32 + all32) * 4 <= 80 * 1024) ? 1 : 0;
        lds_bytes = (fixed32 + (a.resident ? all32 : 2 * max32)) * 4;
        a.coord_table = 0;
        if (!a.pos) {
            const long long tbl = 4LL * ((long long)a.res0 + a.res1 + a.res2);
            const long long cap = a.resident ? 80 * 1024 : 160 * 1024;
            if (lds_bytes + tbl <= cap) { a.coord_table = 1; lds_bytes += (int)tbl; }
        }
        a.waves = (!a.resident && (n + 255) / 256 >= num_cus()) ? 8 : 4;
        a.nbatches = (n + 255) / 256 * (8 / a.waves);
        grid = (a.resident ? 2LL : 1LL) * num_cus();
        if (grid > a.nbatches) grid = a.nbatches;
    }
    switch (p.CH) {
        case 8: return lfgc_fwd_dispatch_ch8(p.MT, a, lds_bytes, (int)grid, st);
        case 16: return lfgc_fwd_dispatch_ch16(p.MT, a, lds_bytes, (int)grid, st);
        case 24: return lfgc_fwd_dispatch_ch24(p.MT, a, lds_bytes, (int)grid, st);
        case 32: return lfgc_fwd_dispatch_ch32(p.MT, a, lds_bytes, (int)grid, st);
        default: return LFGC_E_UNSUPPORTED;
    }
}

extern "C" int lfgc_forward_bf16(const lfgc_mlp_desc* desc, const lfgc_positions* positions, const float* grid_cl, int D, int H,
                                 int W, const float* packed, int clamp, float* out, float* stash, int32_t* status,
                                 lfgc_stream_t stream) {
    return lfgc_forward_f32(desc, positions, grid_cl, D, H, W, packed, LFGC_PRECISION_F16, clamp, out, stash, status, stream);
}
