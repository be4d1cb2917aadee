// lfgc_wavelet.hip -- 4-tap (db2) separable-bank 3-D wavelet kernels for gfx950.
//   lfgc_idwt_level_f32      replaces wavelet_transform/Torch_Wavelet_Transform.py:91-104 (+ crop :69-73)
//   lfgc_idwt_level_bwd_f32  its adjoint (autograd of the same lines)
//   lfgc_dwt_level_f32       replaces :59-67, :75-89 (init-time encode)
//   lfgc_grid_layout_f32     channel-first <-> channel-last conversion of the dense grid (the sampler and the
//                            gradient scatter work channel-last, the stencils channel-first)
// All of them are HBM/L2-bound byte movers (no contraction worth an MFMA).  Both stencils use the same shape:
// one thread owns one coarse cell of one channel, issues its 64 loads up front (predicated by clamped index and a
// zeroing select, never by a branch, so they are all in flight together), then forms 8 results with 64 FMAs each
// against filter taps that are wave-uniform scalar loads; lanes run along the last spatial axis.
#include "lfgc_common.h"

namespace {

struct IdwtArgs {
    const float* lll;   // (C, d0,d1,d2)
    const float* hf;    // (C, 7, d0,d1,d2)
    const float* filt;  // (8,4,4,4)
    float* out;         // (C, t0,t1,t2)
    int C, d0, d1, d2, t0, t1, t2, o0, o1, o2;   // o = crop offset floor((2d+2-t)/2)
    // DROP build only: the pruning layers' per-coefficient factors, shared by all channels
    const float* mul_l; // (d0,d1,d2) or NULL
    const float* mul_h; // (7, d0,d1,d2) or NULL
    float thr_l, thr_h; // NaN: value = x * m;  else masked straight-through: value = (x*(m>=thr) - x*m) + x*m
};

// One coefficient through its drop layer (model/Smallify_Dropout.py:57, model/Variational_Dropout_Layer.py:109,
// model/Straight_Through_Dropout.py:28 and :58 -- the latter op for op, so the value is the reference's bit for bit).
__device__ __forceinline__ float drop_value(float x, float m, float thr, bool ste) {
    if (!ste) return __fmul_rn(x, m);
    const float hard = m >= thr ? 1.0f : 0.0f;
    const float soft = __fmul_rn(x, m);
    return __fadd_rn(__fsub_rn(__fmul_rn(x, hard), soft), soft);
}

// Synthesis: out_full[o] = sum_{s,t} in[s][i] F_s[t], o = 2 i + t per axis.  Thread = cell jj in [0,d] per axis:
// it produces the 2x2x2 outputs o = 2 jj + p from the cells i = jj - e (e in {0,1}) with taps t = p + 2 e.
template <bool DROP>
__global__ __launch_bounds__(256) void idwt_level_kernel(const IdwtArgs a) {
    // filter bank re-laid [tap][band] in LDS: the 8 bands of one tap are two broadcast ds_read_b128
    // (left in global memory hipcc fetches every tap with a per-lane vector load: 512 extra loads per thread)
    __shared__ __attribute__((aligned(16))) float s_f[512];
    for (int i = threadIdx.x; i < 512; i += 256) s_f[(i & 63) * 8 + (i >> 6)] = a.filt[i];
    __syncthreads();
    const int n0 = a.d0 + 1, n1 = a.d1 + 1, n2 = a.d2 + 1;
    const long long total = (long long)a.C * n0 * n1 * n2;
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    // one cell per thread, no grid-stride loop: a loop makes the LDS filter reads loop-invariant and hipcc then
    // hoists all 512 taps into VGPRs (256 VGPRs + scratch, occupancy 1)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < total) {
        const int jx = (int)(idx % n2);
        long long r = idx / n2;
        const int jy = (int)(r % n1); r /= n1;
        const int jz = (int)(r % n0);
        const int c = (int)(r / n0);
        const float* in_l = a.lll + (long long)c * dvol;
        const float* in_h = a.hf + (long long)c * 7 * dvol;
        float v[8][8];                                       // [e = ez*4+ey*2+ex][band]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int iz = jz - (e >> 2), iy = jy - ((e >> 1) & 1), ix = jx - (e & 1);
            const bool ok = iz >= 0 && iz < a.d0 && iy >= 0 && iy < a.d1 && ix >= 0 && ix < a.d2;
            const long long sp = ((long long)min(max(iz, 0), a.d0 - 1) * a.d1 + min(max(iy, 0), a.d1 - 1)) * a.d2 +
                                 min(max(ix, 0), a.d2 - 1);
            float l = in_l[sp];
            if (DROP && a.mul_l) l = drop_value(l, a.mul_l[sp], a.thr_l, a.thr_l == a.thr_l);
            v[e][0] = ok ? l : 0.0f;
#pragma unroll
            for (int s = 1; s < 8; ++s) {
                float h = in_h[(long long)(s - 1) * dvol + sp];
                if (DROP && a.mul_h) h = drop_value(h, a.mul_h[(long long)(s - 1) * dvol + sp], a.thr_h, a.thr_h == a.thr_h);
                v[e][s] = ok ? h : 0.0f;
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pz = p >> 2, py = (p >> 1) & 1, px = p & 1;
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int tap = ((pz + 2 * (e >> 2)) * 4 + (py + 2 * ((e >> 1) & 1))) * 4 + (px + 2 * (e & 1));
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(s_f + tap * 8);
                const f32x4 f1 = *reinterpret_cast<const f32x4*>(s_f + tap * 8 + 4);
                acc = __builtin_fmaf(v[e][0], f0.x, acc); acc = __builtin_fmaf(v[e][1], f0.y, acc);
                acc = __builtin_fmaf(v[e][2], f0.z, acc); acc = __builtin_fmaf(v[e][3], f0.w, acc);
                acc = __builtin_fmaf(v[e][4], f1.x, acc); acc = __builtin_fmaf(v[e][5], f1.y, acc);
                acc = __builtin_fmaf(v[e][6], f1.z, acc); acc = __builtin_fmaf(v[e][7], f1.w, acc);
            }
            const int oz = 2 * jz + pz - a.o0, oy = 2 * jy + py - a.o1, ox = 2 * jx + px - a.o2;
            if (oz >= 0 && oz < a.t0 && oy >= 0 && oy < a.t1 && ox >= 0 && ox < a.t2)
                a.out[(((long long)c * a.t0 + oz) * a.t1 + oy) * a.t2 + ox] = acc;
        }
    }
}

// Analysis-form kernel shared by the IDWT adjoint and the forward DWT:
//   band_s[c][i] = sum_t src_full[c][2 i + t] * F_s[t],  src_full[u] = src[u - lo] (0 outside [0, n))
struct AnalysisArgs {
    const float* src;      // (C, n0,n1,n2)
    const float* filt;
    float* band0;          // band 0 of channel c at band0 + c * cstride0
    float* bandh;          // band s>=1 of channel c at bandh + c * cstrideh + (s-1) * dvol
    long long cstride0, cstrideh;
    int C, n0, n1, n2, lo0, lo1, lo2, d0, d1, d2;
};

__global__ __launch_bounds__(256) void analysis_kernel(const AnalysisArgs a) {
    __shared__ __attribute__((aligned(16))) float s_f[512];      // [tap][band], see idwt_level_kernel
    for (int i = threadIdx.x; i < 512; i += 256) s_f[(i & 63) * 8 + (i >> 6)] = a.filt[i];
    __syncthreads();
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    const long long nvol = (long long)a.n0 * a.n1 * a.n2;
    const long long total = (long long)a.C * dvol;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // one cell per thread (see idwt_level_kernel)
    if (idx < total) {
        const int ix = (int)(idx % a.d2);
        long long r = idx / a.d2;
        const int iy = (int)(r % a.d1); r /= a.d1;
        const int iz = (int)(r % a.d0);
        const int c = (int)(r / a.d0);
        const float* src = a.src + (long long)c * nvol;
        float acc[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = 0.0f;
#pragma unroll 1                      // a rolled z-tap loop keeps the live filter taps to one 16-tap plane
        for (int tz = 0; tz < 4; ++tz) {
            const int uz = 2 * iz + tz - a.lo0;
            const bool okz = uz >= 0 && uz < a.n0;
            const int cz = min(max(uz, 0), a.n0 - 1);
            float v[16];
#pragma unroll
            for (int tyx = 0; tyx < 16; ++tyx) {
                const int uy = 2 * iy + (tyx >> 2) - a.lo1, ux = 2 * ix + (tyx & 3) - a.lo2;
                const bool ok = okz && uy >= 0 && uy < a.n1 && ux >= 0 && ux < a.n2;
                const float x = src[((long long)cz * a.n1 + min(max(uy, 0), a.n1 - 1)) * a.n2 + min(max(ux, 0), a.n2 - 1)];
                v[tyx] = ok ? x : 0.0f;
            }
#pragma unroll
            for (int tyx = 0; tyx < 16; ++tyx) {
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(s_f + (tz * 16 + tyx) * 8);
                const f32x4 f1 = *reinterpret_cast<const f32x4*>(s_f + (tz * 16 + tyx) * 8 + 4);
                acc[0] = __builtin_fmaf(v[tyx], f0.x, acc[0]); acc[1] = __builtin_fmaf(v[tyx], f0.y, acc[1]);
                acc[2] = __builtin_fmaf(v[tyx], f0.z, acc[2]); acc[3] = __builtin_fmaf(v[tyx], f0.w, acc[3]);
                acc[4] = __builtin_fmaf(v[tyx], f1.x, acc[4]); acc[5] = __builtin_fmaf(v[tyx], f1.y, acc[5]);
                acc[6] = __builtin_fmaf(v[tyx], f1.z, acc[6]); acc[7] = __builtin_fmaf(v[tyx], f1.w, acc[7]);
            }
        }
        const long long sp_out = ((long long)iz * a.d1 + iy) * a.d2 + ix;
        a.band0[(long long)c * a.cstride0 + sp_out] = acc[0];
#pragma unroll
        for (int s = 1; s < 8; ++s) a.bandh[(long long)c * a.cstrideh + (long long)(s - 1) * dvol + sp_out] = acc[s];
    }
}

// Adjoint of the IDWT level WITH drop factors (autograd of "coefficients * factor -> conv_transpose3d"):
//   adj_s[c][i]  = analysis of d_out (as analysis_kernel),   d_coef_s[c][i] = adj_s[c][i] * m_s[i],
//   d_m_s[i]     = sum_c adj_s[c][i] * coef_s[c][i]          (the factor is shared by all channels).
// Block = 4 channel slots (one wave each) x 64 cells; a wave walks the channels c = slot, slot+4, ... of its 64 cells,
// keeps the 8 partial d_m in registers, and the 4 slots are combined through LDS in a fixed order: no atomics, the
// result is bitwise repeatable.  The channel loop and the z-tap loop stay rolled (see idwt_level_kernel on LICM).
struct AdjointDropArgs {
    const float* src;      // d_out (C, n0,n1,n2)
    const float* filt;
    const float* lll;      // forward inputs: (C, d0,d1,d2)
    const float* hf;       //                 (C, 7, d0,d1,d2)
    const float* mul_l;    // (d0,d1,d2) or NULL
    const float* mul_h;    // (7, d0,d1,d2) or NULL
    float* d_lll;          // (C, d0,d1,d2)
    float* d_hf;           // (C, 7, d0,d1,d2)
    float* d_mul_l;        // (d0,d1,d2) or NULL
    float* d_mul_h;        // (7, d0,d1,d2) or NULL
    int C, n0, n1, n2, lo0, lo1, lo2, d0, d1, d2;
};

__global__ __launch_bounds__(256) void adjoint_drop_kernel(const AdjointDropArgs a) {
    __shared__ __attribute__((aligned(16))) float s_f[512];      // [tap][band]
    __shared__ float s_red[3][8][64];                            // partial d_m of slots 1..3
    for (int i = threadIdx.x; i < 512; i += 256) s_f[(i & 63) * 8 + (i >> 6)] = a.filt[i];
    __syncthreads();
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    const long long nvol = (long long)a.n0 * a.n1 * a.n2;
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const long long cell = (long long)blockIdx.x * 64 + lane;
    const bool valid = cell < dvol;
    const long long cc = valid ? cell : dvol - 1;
    const int ix = (int)(cc % a.d2);
    const long long r = cc / a.d2;
    const int iy = (int)(r % a.d1), iz = (int)(r / a.d1);
    float m[8], dm[8];
    m[0] = a.mul_l ? a.mul_l[cc] : 1.0f;
#pragma unroll
    for (int s = 1; s < 8; ++s) m[s] = a.mul_h ? a.mul_h[(long long)(s - 1) * dvol + cc] : 1.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) dm[s] = 0.0f;
#pragma unroll 1
    for (int c = slot; c < a.C; c += 4) {
        const float* src = a.src + (long long)c * nvol;
        float acc[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = 0.0f;
#pragma unroll 1
        for (int tz = 0; tz < 4; ++tz) {
            const int uz = 2 * iz + tz - a.lo0;
            const bool okz = uz >= 0 && uz < a.n0;
            const int cz = min(max(uz, 0), a.n0 - 1);
            float v[16];
#pragma unroll
            for (int tyx = 0; tyx < 16; ++tyx) {
                const int uy = 2 * iy + (tyx >> 2) - a.lo1, ux = 2 * ix + (tyx & 3) - a.lo2;
                const bool ok = okz && uy >= 0 && uy < a.n1 && ux >= 0 && ux < a.n2;
                const float x = src[((long long)cz * a.n1 + min(max(uy, 0), a.n1 - 1)) * a.n2 + min(max(ux, 0), a.n2 - 1)];
                v[tyx] = ok ? x : 0.0f;
            }
#pragma unroll
            for (int tyx = 0; tyx < 16; ++tyx) {
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(s_f + (tz * 16 + tyx) * 8);
                const f32x4 f1 = *reinterpret_cast<const f32x4*>(s_f + (tz * 16 + tyx) * 8 + 4);
                acc[0] = __builtin_fmaf(v[tyx], f0.x, acc[0]); acc[1] = __builtin_fmaf(v[tyx], f0.y, acc[1]);
                acc[2] = __builtin_fmaf(v[tyx], f0.z, acc[2]); acc[3] = __builtin_fmaf(v[tyx], f0.w, acc[3]);
                acc[4] = __builtin_fmaf(v[tyx], f1.x, acc[4]); acc[5] = __builtin_fmaf(v[tyx], f1.y, acc[5]);
                acc[6] = __builtin_fmaf(v[tyx], f1.z, acc[6]); acc[7] = __builtin_fmaf(v[tyx], f1.w, acc[7]);
            }
        }
        if (valid) {
            const long long ol = (long long)c * dvol + cc;
            if (a.d_mul_l) dm[0] = __builtin_fmaf(acc[0], a.lll[ol], dm[0]);
            a.d_lll[ol] = a.mul_l ? acc[0] * m[0] : acc[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) {
                const long long oh = ((long long)c * 7 + (s - 1)) * dvol + cc;
                if (a.d_mul_h) dm[s] = __builtin_fmaf(acc[s], a.hf[oh], dm[s]);
                a.d_hf[oh] = a.mul_h ? acc[s] * m[s] : acc[s];
            }
        }
    }
    if (slot > 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) s_red[slot - 1][s][lane] = dm[s];
    }
    __syncthreads();
    if (slot == 0 && valid) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float t = (dm[s] + s_red[0][s][lane]) + (s_red[1][s][lane] + s_red[2][s][lane]);
            if (s == 0) { if (a.d_mul_l) a.d_mul_l[cc] = t; }
            else if (a.d_mul_h) a.d_mul_h[(long long)(s - 1) * dvol + cc] = t;
        }
    }
}

// (C, V) <-> (V, Cs) through a 32(channel) x 64(voxel) LDS tile: both sides move whole 128/256-byte rows.
__global__ __launch_bounds__(256) void first_to_last_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int C, long long V, int cs) {
    __shared__ float tile[32][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
    const long long v0 = (long long)blockIdx.x * 64;
    for (int c0 = 0; c0 < cs; c0 += 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + ty * 8 + k;
            const long long v = v0 + tx;
            tile[ty * 8 + k][tx] = (c < C && v < V) ? src[(long long)c * V + v] : 0.0f;
        }
        __syncthreads();
        const int cc = threadIdx.x & 31, vv = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long v = v0 + vv * 8 + k;
            if (v < V && c0 + cc < cs) dst[v * cs + c0 + cc] = tile[cc][vv * 8 + k];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void last_to_first_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int C, long long V, int cs) {
    __shared__ float tile[32][65];
    const long long v0 = (long long)blockIdx.x * 64;
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int cc = threadIdx.x & 31, vv = threadIdx.x >> 5;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long v = v0 + vv * 8 + k;
            tile[cc][vv * 8 + k] = (v < V && c0 + cc < cs) ? src[v * cs + c0 + cc] : 0.0f;
        }
        __syncthreads();
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + ty * 8 + k;
            const long long v = v0 + tx;
            if (c < C && v < V) dst[(long long)c * V + v] = tile[ty * 8 + k][tx];
        }
        __syncthreads();
    }
}

inline unsigned grid_for(long long total, int block = 256) {
    long long g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g);
}

inline int check_level(const void* a, const void* b, const void* c, const void* d, int C, int d0, int d1, int d2,
                       int t0, int t1, int t2) {
    if (!a || !b || !c || !d) return LFGC_E_NULL;
    if (C < 1 || d0 < 1 || d1 < 1 || d2 < 1 || t0 < 1 || t1 < 1 || t2 < 1) return LFGC_E_SHAPE;
    if (t0 > 2 * d0 + 2 || t1 > 2 * d1 + 2 || t2 > 2 * d2 + 2) return LFGC_E_SHAPE;
    return LFGC_OK;
}

}  // namespace

extern "C" int lfgc_idwt_level_f32(const float* lll, const float* hf, const float* filter_rev, float* out,
                                   int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(lll, hf, filter_rev, out, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    IdwtArgs a;
    a.lll = lll; a.hf = hf; a.filt = filter_rev; a.out = out;
    a.C = C; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
    a.o0 = (2 * d0 + 2 - t0) / 2; a.o1 = (2 * d1 + 2 - t1) / 2; a.o2 = (2 * d2 + 2 - t2) / 2;
    const long long total = (long long)C * (d0 + 1) * (d1 + 1) * (d2 + 1);
    a.mul_l = nullptr; a.mul_h = nullptr; a.thr_l = a.thr_h = 0.0f;
    hipLaunchKernelGGL(idwt_level_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_idwt_level_drop_f32(const float* lll, const float* hf, const float* mul_lll, float thr_lll,
                                        const float* mul_hf, float thr_hf, const float* filter_rev, float* out,
                                        int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(lll, hf, filter_rev, out, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    IdwtArgs a;
    a.lll = lll; a.hf = hf; a.filt = filter_rev; a.out = out;
    a.C = C; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
    a.o0 = (2 * d0 + 2 - t0) / 2; a.o1 = (2 * d1 + 2 - t1) / 2; a.o2 = (2 * d2 + 2 - t2) / 2;
    a.mul_l = mul_lll; a.mul_h = mul_hf; a.thr_l = thr_lll; a.thr_h = thr_hf;
    const long long total = (long long)C * (d0 + 1) * (d1 + 1) * (d2 + 1);
    if (mul_lll || mul_hf)
        hipLaunchKernelGGL(idwt_level_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(idwt_level_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_idwt_level_bwd_f32(const float* d_out, const float* filter_rev, float* d_lll, float* d_hf,
                                       int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(d_out, filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    AnalysisArgs a;
    a.src = d_out; a.filt = filter_rev; a.band0 = d_lll; a.bandh = d_hf;
    const long long dvol = (long long)d0 * d1 * d2;
    a.cstride0 = dvol; a.cstrideh = 7 * dvol;
    a.C = C; a.n0 = t0; a.n1 = t1; a.n2 = t2;
    a.lo0 = (2 * d0 + 2 - t0) / 2; a.lo1 = (2 * d1 + 2 - t1) / 2; a.lo2 = (2 * d2 + 2 - t2) / 2;
    a.d0 = d0; a.d1 = d1; a.d2 = d2;
    hipLaunchKernelGGL(analysis_kernel, dim3(grid_for((long long)C * dvol)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_idwt_level_drop_bwd_f32(const float* d_out, const float* filter_rev, const float* lll, const float* hf,
                                            const float* mul_lll, const float* mul_hf, float* d_lll, float* d_hf,
                                            float* d_mul_lll, float* d_mul_hf, int C, int d0, int d1, int d2,
                                            int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(d_out, filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    if ((d_mul_lll && (!mul_lll || !lll)) || (d_mul_hf && (!mul_hf || !hf))) return LFGC_E_NULL;
    AdjointDropArgs a;
    a.src = d_out; a.filt = filter_rev; a.lll = lll; a.hf = hf; a.mul_l = mul_lll; a.mul_h = mul_hf;
    a.d_lll = d_lll; a.d_hf = d_hf; a.d_mul_l = d_mul_lll; a.d_mul_h = d_mul_hf;
    a.C = C; a.n0 = t0; a.n1 = t1; a.n2 = t2;
    a.lo0 = (2 * d0 + 2 - t0) / 2; a.lo1 = (2 * d1 + 2 - t1) / 2; a.lo2 = (2 * d2 + 2 - t2) / 2;
    a.d0 = d0; a.d1 = d1; a.d2 = d2;
    const long long dvol = (long long)d0 * d1 * d2;
    hipLaunchKernelGGL(adjoint_drop_kernel, dim3(grid_for(dvol, 64)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_dwt_level_f32(const float* in, const float* filter_fwd, float* out,
                                  int C, int n0, int n1, int n2, lfgc_stream_t stream) {
    if (!in || !filter_fwd || !out) return LFGC_E_NULL;
    if (C < 1 || n0 < 1 || n1 < 1 || n2 < 1) return LFGC_E_SHAPE;
    // _get_padding_size (Torch_Wavelet_Transform.py:59-63): F.pad slots are (last axis lo, hi, ..., first axis
    // lo, hi) while is_odd is indexed first axis first -> the odd bit of axis a pads axis 2-a.
    const int hi0 = 2 + (n2 & 1), hi1 = 2 + (n1 & 1), hi2 = 2 + (n0 & 1);
    AnalysisArgs a;
    a.src = in; a.filt = filter_fwd;
    a.d0 = (n0 + 2 + hi0 - 4) / 2 + 1; a.d1 = (n1 + 2 + hi1 - 4) / 2 + 1; a.d2 = (n2 + 2 + hi2 - 4) / 2 + 1;
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    a.band0 = out; a.bandh = out + dvol;
    a.cstride0 = 8 * dvol; a.cstrideh = 8 * dvol;
    a.C = C; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.lo0 = 2; a.lo1 = 2; a.lo2 = 2;
    hipLaunchKernelGGL(analysis_kernel, dim3(grid_for((long long)C * dvol)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_grid_layout_f32(const float* src, float* dst, int C, int64_t voxels, int channel_stride,
                                    int to_channel_last, lfgc_stream_t stream) {
    if (!src || !dst) return LFGC_E_NULL;
    if (C < 1 || voxels < 1 || channel_stride < C) return LFGC_E_SHAPE;
    const unsigned g = (unsigned)((voxels + 63) / 64);
    if (to_channel_last)
        hipLaunchKernelGGL(first_to_last_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, dst, C, (long long)voxels, channel_stride);
    else
        hipLaunchKernelGGL(last_to_first_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, dst, C, (long long)voxels, channel_stride);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}
