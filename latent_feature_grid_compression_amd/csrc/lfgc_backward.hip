// lfgc_backward.hip -- C-ABI entry for the backward of the fused path: checks, workspace carving, dispatch.
#include <cstdlib>
#include "lfgc_backward.h"

int lfgc_bwd_dispatch_ch8(int, const LfgcBwdArgs&, const LfgcWgradArgs&, int, int, int, int, int, hipStream_t);
int lfgc_bwd_dispatch_ch16(int, const LfgcBwdArgs&, const LfgcWgradArgs&, int, int, int, int, int, hipStream_t);
int lfgc_bwd_dispatch_ch24(int, const LfgcBwdArgs&, const LfgcWgradArgs&, int, int, int, int, int, hipStream_t);
int lfgc_bwd_dispatch_ch32(int, const LfgcBwdArgs&, const LfgcWgradArgs&, int, int, int, int, int, hipStream_t);

#ifdef LFGC_STAMPS
static unsigned long long* g_bwd_stamps = nullptr;
// Diagnostics builds only (tools/phase_stamps.py bwd): device buffer of 20 counters per wave slot of the data kernel.
extern "C" void lfgc_debug_set_bwd_stamp_buffer(void* p) { g_bwd_stamps = reinterpret_cast<unsigned long long*>(p); }
#endif

namespace {
#ifndef LFGC_MAX_SLABS
#define LFGC_MAX_SLABS 256
#endif
const int kMaxSlabs = LFGC_MAX_SLABS;   // workgroups of the weight-gradient kernel (one partial slab each)

struct Carve {
    long long ntiles, nbatches;
    int nslabs;                // partial slabs = tile groups of the weight-gradient kernel
    int roles;                 // workgroups per tile group (LfgcWgradArgs::roles)
    long long dstash_floats, slab_floats_total, dscale_floats, dfeat_floats;
};

Carve carve(const LfgcPlan& p, long long n) {
    Carve c;
    c.nbatches = (n + 255) / 256;                      // whole 256-sample groups, like the forward's stash
    c.ntiles = c.nbatches * 8;
    c.nslabs = (int)(c.ntiles < kMaxSlabs ? c.ntiles : kMaxSlabs);
    if (c.nslabs < 1) c.nslabs = 1;
    // enough tiles: one workgroup per (tile group, layer) instead of per tile group -- the same kMaxSlabs workgroups read
    // the same operands but leave L times fewer slabs (cfg-3 step: 69 -> 17 MB written, and read again by the reduction)
    c.roles = 1;
    if (c.ntiles >= 2LL * kMaxSlabs && p.L > 1 && !getenv("LFGC_WGRAD_NO_SPLIT")) {
        c.roles = p.L;
        c.nslabs = kMaxSlabs / p.L;
    }
    c.dstash_floats = c.ntiles * 64LL * (p.L * 16 * p.MT);
    c.slab_floats_total = (long long)c.nslabs * lfgc_slab_floats(p);
    c.dscale_floats = (c.ntiles * p.L + 3) / 4 * 4;     // one power-of-two scale per (tile, layer), f16 builds
    c.dfeat_floats = c.ntiles * 32 * p.CH;              // feature gradients for the deferred scatter (small batches)
    return c;
}
}  // namespace

extern "C" int64_t lfgc_backward_workspace_bytes(const lfgc_mlp_desc* desc, int64_t n_samples) {
    if (!lfgc_mlp_supported(desc)) return LFGC_E_UNSUPPORTED;
    if (n_samples < 0) return LFGC_E_SHAPE;
    const LfgcPlan p = lfgc_make_plan(desc->grid_channels, desc->hidden, desc->num_layers, desc->n_freqs);
    const Carve c = carve(p, n_samples);
    return (c.dstash_floats + c.slab_floats_total + c.dscale_floats + c.dfeat_floats) * 4;
}

extern "C" int lfgc_backward_f32(const lfgc_mlp_desc* desc, const lfgc_positions* positions,
                                 const float* grid_cl, int D, int H, int W,
                                 const float* packed, int precision, const float* stash, const float* d_out,
                                 float* d_grid_cl, float* const* d_weights, float* const* d_biases, float* d_pos,
                                 void* workspace, int64_t workspace_bytes, lfgc_stream_t stream) {
    if (!desc || !positions || !grid_cl || !packed || !stash || !d_out || !d_grid_cl || !d_weights || !d_biases)
        return LFGC_E_NULL;
    if (!lfgc_mlp_supported(desc)) return LFGC_E_UNSUPPORTED;
    if (precision != LFGC_PRECISION_F32 && precision != LFGC_PRECISION_F16X2 && precision != LFGC_PRECISION_F16) return LFGC_E_UNSUPPORTED;
    if (!positions->pos) return LFGC_E_NULL;            // backward runs on explicit positions only
    if (positions->n < 0 || D < 1 || H < 1 || W < 1) return LFGC_E_SHAPE;
    if ((((uintptr_t)grid_cl) | ((uintptr_t)packed) | ((uintptr_t)stash) | ((uintptr_t)d_grid_cl) | ((uintptr_t)workspace)) & 15)
        return LFGC_E_ALIGN;
    const LfgcPlan p = lfgc_make_plan(desc->grid_channels, desc->hidden, desc->num_layers, desc->n_freqs);
    const long long n = positions->n;
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l <= p.L; ++l)
        if (!d_weights[l] || !d_biases[l]) return LFGC_E_NULL;
    if (n == 0) {                                       // gradients of an empty batch are zero
        const int K0 = p.E + p.C;
        for (int l = 0; l <= p.L; ++l) {
            const size_t wn = (l == 0) ? (size_t)p.H * K0 : (l == p.L ? (size_t)p.H : (size_t)p.H * p.H);
            const size_t bn = (l == p.L) ? 1 : (size_t)p.H;
            hipError_t e = hipMemsetAsync(d_weights[l], 0, wn * 4, st);
            if (e == hipSuccess) e = hipMemsetAsync(d_biases[l], 0, bn * 4, st);
            if (e != hipSuccess) return (int)e;
        }
        return LFGC_OK;
    }
    const Carve c = carve(p, n);
    if (!workspace || workspace_bytes < (c.dstash_floats + c.slab_floats_total + c.dscale_floats + c.dfeat_floats) * 4) return LFGC_E_WORKSPACE;
    float* dstash = reinterpret_cast<float*>(workspace);
    float* slabs = dstash + c.dstash_floats;
    float* dscale = slabs + c.slab_floats_total;
    float* dfeat = dscale + c.dscale_floats;

    LfgcBwdArgs a;
    a.pos = positions->pos; a.n = n;
    a.grid = grid_cl; a.D = D; a.H = H; a.W = W; a.Cs = p.CH;
    a.packed = packed; a.L = p.L; a.stash = stash; a.d_out = d_out;
    a.dstash = dstash; a.dscale = dscale; a.d_grid = d_grid_cl; a.d_pos = d_pos;
    a.stamps = nullptr;
#ifdef LFGC_STAMPS
    a.stamps = g_bwd_stamps;
#endif

    LfgcWgradArgs w;
    w.stash = stash; w.dstash = dstash; w.d_out = d_out; w.n = n; w.ntiles = c.ntiles; w.L = p.L;
    w.slabs = slabs; w.slab_floats = lfgc_slab_floats(p); w.roles = c.roles;
    w.dscale = precision == LFGC_PRECISION_F32 ? nullptr : dscale;     // f16 builds: f16-split contraction (lfgc_backward.h)

    const int cus = lfgc_num_cus();       // per device
    // data kernel: one workgroup per CU, 8 waves once every CU gets a 256-sample batch, else 4 (tiles beyond the
    // last whole 128-sample group are never touched: the stash covers whole 256-sample groups, lfgc_stash_bytes)
    const int waves = ((n + 255) / 256 >= cus) ? 8 : 4;
    // Feature-gradient scatter: inside the data kernel.  LFGC_SCATTER=deferred (diagnostics) moves the float atomics into
    // a kernel of their own at full occupancy: measured at the cfg-3 train step, the data kernel drops from 78 to 54 us and
    // the scatter kernel takes 29 us -- 8.4 M device-scope float adds on cold lines cost that much either way (the
    // in-kernel phase stamps' 40 % "scatter" share is their latency, not an occupancy problem), so the step does not move.
    {
        const char* env = getenv("LFGC_SCATTER");
        a.dfeat = (env && env[0] == 'd') ? dfeat : nullptr;
    }
    a.nbatches = c.nbatches * (8 / waves);              // same tile range as the forward wrote
    const int tb0 = p.K0R * p.ST, tb1 = p.HP * p.ST, sc = waves * 32 * (p.CH + 4 + 16);
    int slot = tb0 > tb1 ? tb0 : tb1;
    if (sc > slot) slot = sc;
    const int lds_bytes = (p.HP + 4 + 8 + 2 * slot) * 4;
    long long grid_data = cus;
    if (grid_data > a.nbatches) grid_data = a.nbatches;

    int rc;
    switch (p.CH) {
        case 8: rc = lfgc_bwd_dispatch_ch8(p.MT, a, w, waves, precision, lds_bytes, (int)grid_data, c.nslabs * c.roles, st); break;
        case 16: rc = lfgc_bwd_dispatch_ch16(p.MT, a, w, waves, precision, lds_bytes, (int)grid_data, c.nslabs * c.roles, st); break;
        case 24: rc = lfgc_bwd_dispatch_ch24(p.MT, a, w, waves, precision, lds_bytes, (int)grid_data, c.nslabs * c.roles, st); break;
        case 32: rc = lfgc_bwd_dispatch_ch32(p.MT, a, w, waves, precision, lds_bytes, (int)grid_data, c.nslabs * c.roles, st); break;
        default: return LFGC_E_UNSUPPORTED;
    }
    if (rc != LFGC_OK) return rc;

    LfgcReduceArgs r;
    r.slabs = slabs; r.nslabs = c.nslabs; r.slab_floats = w.slab_floats; r.plan = p;
    for (int l = 0; l <= p.L; ++l) { r.dw[l] = d_weights[l]; r.db[l] = d_biases[l]; }
    for (int i = 0; i < 64; ++i) r.col_of_src[i] = 0;
    for (int cl = 0; cl < p.K0P; ++cl) {
        const int src = lfgc_layer0_src_col(p, cl);
        if (src >= 0) r.col_of_src[src] = cl;
    }
    const int K0 = p.E + p.C;
    const int total = p.H * K0 + p.H + (p.L - 1) * (p.H * p.H + p.H) + p.H + 1;
    const int g = (total + 63) / 64;
    hipLaunchKernelGGL(lfgc_bwd_reduce_kernel, dim3(g), dim3(256), 0, st, r);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_backward_bf16(const lfgc_mlp_desc* desc, const lfgc_positions* positions, const float* grid_cl, int D, int H,
                                  int W, const float* packed, const float* stash, const float* d_out, float* d_grid_cl,
                                  float* const* d_weights, float* const* d_biases, float* d_pos,
                                  void* workspace, int64_t workspace_bytes, lfgc_stream_t stream) {
    return lfgc_backward_f32(desc, positions, grid_cl, D, H, W, packed, LFGC_PRECISION_F16, stash, d_out, d_grid_cl, d_weights,
                             d_biases, d_pos, workspace, workspace_bytes, stream);
}
