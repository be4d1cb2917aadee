// lfgc_wavelet.hip -- 4-tap (db2) separable-bank 3-D wavelet kernels for gfx950.
//   lfgc_idwt_level_f32           replaces wavelet_transform/Torch_Wavelet_Transform.py:91-104 (+ crop :69-73)
//   lfgc_idwt_level_bwd_f32       its adjoint (autograd of the same lines)
//   lfgc_idwt_level_drop_f32/_bwd the same two with the pruning layers' per-coefficient factors folded in
//   lfgc_dwt_level_f32            replaces :59-67, :75-89 (init-time encode)
//   lfgc_grid_layout_f32          channel-first <-> channel-last conversion of the dense grid (the sampler and the
//                                 gradient scatter work channel-last, the stencils channel-first)
// All of them are byte movers with a 64-FMA-per-output stencil (no contraction worth an MFMA).  Both stencils share
// one shape: a workgroup owns 2 z-slices of a run of consecutive cells of the flattened (y,x) plane of ONE channel, copies
// the input neighbourhood of those cells into LDS as plain contiguous chunks of the source rows (every source value
// is fetched from memory once per workgroup instead of once per reading thread: the per-thread version was bound by
// the 64 vector-load instructions each thread issued), then every thread forms its 8 results from LDS with filter
// taps that are broadcast LDS reads.  Boundary handling is a select on the LDS value, never a branch.
#include "lfgc_common.h"

#ifndef LFGC_WAVELET_ABLATE
#define LFGC_WAVELET_ABLATE 0          // diagnostics (tools/ablate_wavelet.py): 1 no output stores, 2 no arithmetic, 4 no staging loads
#endif

namespace {

struct IdwtArgs {
    const float* lll;   // (C, d0,d1,d2)
    const float* hf;    // (C, 7, d0,d1,d2)
    const float* filt;  // (8,4,4,4)
    float* out;         // (C, t0,t1,t2)
    int C, d0, d1, d2, t0, t1, t2, o0, o1, o2;   // o = crop offset floor((2d+2-t)/2)
    int len;            // plane offsets per staged z-plane
    int zchunk;         // sliding-window kernel: output z-slices per workgroup
    float taps[8];      // SEP build: the 1-D bank the filter is the outer product of: [low | high][tap]
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

constexpr int kTileCells = 128;      // analysis: cells of the flattened (y,x) plane per z-slice of a workgroup
constexpr int kFwdCells = 256;       // synthesis: plane cells per workgroup (one per thread, both z-slices each)

// Synthesis: out_full[o] = sum_{s,t} in[s][i] F_s[t], o = 2 i + t per axis.  Thread = cell jj in [0,d] per axis:
// it produces the 2x2x2 outputs o = 2 jj + p from the cells i = jj - e (e in {0,1}) with taps t = p + 2 e.
// LDS: [512 filter taps as [tap][band]] [3 z-planes (jz0 - 1 + zl)][len plane offsets][12 floats: the 8 bands of that
// cell + pad]; element k is plane offset chunk0 + k, chunk0 = (first cell row - 1) * d2 - 1.  The 48-byte records make
// both the two ds_write_b128 of the staging and the two ds_read_b128 per neighbour cell conflict-free (stride 4 * 3
// dwords).  Offsets outside the plane (and z-planes outside the level) are staged as zeros, so only the x range of a
// neighbour needs a select.
constexpr int kRec = 12;

template <bool DROP, bool SEP>
__global__ __launch_bounds__(256) void idwt_level_kernel(const IdwtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_f = s_dyn;
    float* s_v = s_dyn + 512;
    if (!SEP)
        for (int i = threadIdx.x; i < 512; i += 256) s_f[(i & 63) * 8 + (i >> 6)] = a.filt[i];
    const int n0 = a.d0 + 1, n1 = a.d1 + 1, n2 = a.d2 + 1;
    const int plane_cells = n1 * n2;
    const int pt = blockIdx.x, zt = blockIdx.y, c = blockIdx.z;     // 3-D grid: no index divisions
    const int f0 = pt * kFwdCells, jz0 = zt * 2;
    const int chunk0 = (f0 / n2 - 1) * a.d2 - 1;
    const int len = a.len;
    const int dplane = a.d1 * a.d2;
    const int dvol = dplane * a.d0;                       // < 2^31 / 8 (checked on the host)
    const float* lc = a.lll + (long long)c * dvol;
    const float* hc = a.hf + (long long)c * 7 * dvol;
#pragma unroll 1
    for (int kk = threadIdx.x; kk < len; kk += 256) {
        const int off = chunk0 + kk;
        const bool in_plane = off >= 0 && off < dplane;
        float r[3][8];
#pragma unroll
        for (int zl = 0; zl < 3; ++zl) {
            const int iz = jz0 - 1 + zl;
            const bool ok = in_plane && iz >= 0 && iz < a.d0;
            const int o = ok ? iz * dplane + off : 0;
            r[zl][0] = (LFGC_WAVELET_ABLATE & 4) ? (float)o : lc[o];
#pragma unroll
            for (int sb = 1; sb < 8; ++sb) r[zl][sb] = (LFGC_WAVELET_ABLATE & 4) ? (float)(o + sb) : hc[(sb - 1) * dvol + o];
            if (DROP) {
                if (a.mul_l) r[zl][0] = drop_value(r[zl][0], a.mul_l[o], a.thr_l, a.thr_l == a.thr_l);
                if (a.mul_h) {
#pragma unroll
                    for (int sb = 1; sb < 8; ++sb)
                        r[zl][sb] = drop_value(r[zl][sb], a.mul_h[(sb - 1) * dvol + o], a.thr_h, a.thr_h == a.thr_h);
                }
            }
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) r[zl][sb] = ok ? r[zl][sb] : 0.0f;
        }
#pragma unroll
        for (int zl = 0; zl < 3; ++zl) {
            float* rec = s_v + (zl * len + kk) * kRec;
            *reinterpret_cast<f32x4*>(rec) = f32x4{r[zl][0], r[zl][1], r[zl][2], r[zl][3]};
            *reinterpret_cast<f32x4*>(rec + 4) = f32x4{r[zl][4], r[zl][5], r[zl][6], r[zl][7]};
        }
    }
    __syncthreads();
    // thread = one cell of the (y,x) plane, for BOTH z-slices of the workgroup: the index arithmetic, the x-range
    // selects' predicates and the middle z-plane's 4 neighbour records are shared by its two output cells
    const int f = f0 + threadIdx.x;
    const bool in_plane = f < plane_cells;
    const int fc = min(f, plane_cells - 1);
    const int jy = fc / n2, jx = fc - jy * n2;
    const int k00 = jy * a.d2 + jx - chunk0;
    const bool xok0 = jx < a.d2, xok1 = jx >= 1;
    auto load_plane = [&](int zl, float (&P)[4][8]) {       // P[q = ey*2+ex][band] of z-plane zl
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool xok = (q & 1) ? xok1 : xok0;
            const float* rec = s_v + (zl * len + k00 - (q >> 1) * a.d2 - (q & 1)) * kRec;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(rec);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(rec + 4);
            P[q][0] = xok ? lo.x : 0.0f; P[q][1] = xok ? lo.y : 0.0f; P[q][2] = xok ? lo.z : 0.0f; P[q][3] = xok ? lo.w : 0.0f;
            P[q][4] = xok ? hi.x : 0.0f; P[q][5] = xok ? hi.y : 0.0f; P[q][6] = xok ? hi.z : 0.0f; P[q][7] = xok ? hi.w : 0.0f;
        }
    };
    float* outc = a.out + (long long)c * a.t0 * a.t1 * a.t2;
    // one output cell jz from its neighbour planes: E0 = plane iz = jz (ez = 0), E1 = plane iz = jz - 1 (ez = 1)
    auto cell = [&](int jz, const float (&E0)[4][8], const float (&E1)[4][8]) {
        const bool valid = in_plane && jz < n0;
        auto V = [&](int e, int sb) -> float { return (e >> 2) ? E1[e & 3][sb] : E0[e & 3][sb]; };
        auto store = [&](int p, float val) {
            const int oz = 2 * jz + (p >> 2) - a.o0, oy = 2 * jy + ((p >> 1) & 1) - a.o1, ox = 2 * jx + (p & 1) - a.o2;
            if (valid && oz >= 0 && oz < a.t0 && oy >= 0 && oy < a.t1 && ox >= 0 && ox < a.t2) {
                if (!(LFGC_WAVELET_ABLATE & 1) || val == 1.2345e-30f) outc[(oz * a.t1 + oy) * a.t2 + ox] = val;
            }
        };
        if (LFGC_WAVELET_ABLATE & 2) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float t = 0.0f;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) t += V(p, s8);
                store(p, t);
            }
        } else if (SEP) {
            // F_s[tz][ty][tx] = T[sz][tz] T[sy][ty] T[sx][tx]: contract x, then y, then z in registers (224 FMAs
            // instead of 512, no filter traffic).  Tap of output parity p and neighbour e along one axis: t = p + 2 e.
            float X[2][2][2][2][2];                           // [ez][ey][sz][sy][px]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ez = q >> 3, ey = (q >> 2) & 1, sz = (q >> 1) & 1, sy = q & 1;
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    float t = 0.0f;
#pragma unroll
                    for (int ex = 0; ex < 2; ++ex)
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx)
                            t = __builtin_fmaf(V(ez * 4 + ey * 2 + ex, sz * 4 + sy * 2 + sx), a.taps[sx * 4 + px + 2 * ex], t);
                    X[ez][ey][sz][sy][px] = t;
                }
            }
            float Y[2][2][2][2];                              // [ez][sz][py][px]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ez = q >> 3, sz = (q >> 2) & 1, py = (q >> 1) & 1, px = q & 1;
                float t = 0.0f;
#pragma unroll
                for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                    for (int sy = 0; sy < 2; ++sy)
                        t = __builtin_fmaf(X[ez][ey][sz][sy][px], a.taps[sy * 4 + py + 2 * ey], t);
                Y[ez][sz][py][px] = t;
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int pz = p >> 2, py = (p >> 1) & 1, px = p & 1;
                float t = 0.0f;
#pragma unroll
                for (int ez = 0; ez < 2; ++ez)
#pragma unroll
                    for (int sz = 0; sz < 2; ++sz)
                        t = __builtin_fmaf(Y[ez][sz][py][px], a.taps[sz * 4 + pz + 2 * ez], t);
                store(p, t);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int pz = p >> 2, py = (p >> 1) & 1, px = p & 1;
                float acc = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int tap = ((pz + 2 * (e >> 2)) * 4 + (py + 2 * ((e >> 1) & 1))) * 4 + (px + 2 * (e & 1));
                    const f32x4 f0v = *reinterpret_cast<const f32x4*>(s_f + tap * 8);
                    const f32x4 f1v = *reinterpret_cast<const f32x4*>(s_f + tap * 8 + 4);
                    acc = __builtin_fmaf(V(e, 0), f0v.x, acc); acc = __builtin_fmaf(V(e, 1), f0v.y, acc);
                    acc = __builtin_fmaf(V(e, 2), f0v.z, acc); acc = __builtin_fmaf(V(e, 3), f0v.w, acc);
                    acc = __builtin_fmaf(V(e, 4), f1v.x, acc); acc = __builtin_fmaf(V(e, 5), f1v.y, acc);
                    acc = __builtin_fmaf(V(e, 6), f1v.z, acc); acc = __builtin_fmaf(V(e, 7), f1v.w, acc);
                }
                store(p, acc);      // stored per parity: keeps the filter reads of later parities from being hoisted
            }
        }
    };
    float PA[4][8], PB[4][8];
    load_plane(1, PA);
    load_plane(0, PB);
    cell(jz0, PA, PB);              // outputs of cell jz0: planes jz0 (ez = 0) and jz0 - 1 (ez = 1)
    load_plane(2, PB);
    cell(jz0 + 1, PB, PA);          // cell jz0 + 1: planes jz0 + 1 and jz0
}

// Synthesis with a SLIDING WINDOW along z (separable filters): a workgroup keeps its 256 plane cells and walks `zchunk`
// output slices; output slice jz needs the coefficient planes jz and jz - 1 only, so each step stages ONE new plane (the
// tiled kernel above stages 3 planes per 2 slices), into a 2-slot LDS ring with one barrier per step, and the loads of
// plane jz + 1 are issued right after the barrier so that they fly under the arithmetic of slice jz.  A thread keeps the
// 4 neighbour records of plane jz in registers: they are the ez = 1 operands of the next step.
template <bool DROP, int KI>
__global__ __launch_bounds__(256) void idwt_slide_kernel(const IdwtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_v = s_dyn;                                   // ring[2][len][kRec]: a plane is read from LDS only in its own step
    const int n0 = a.d0 + 1, n1 = a.d1 + 1, n2 = a.d2 + 1;
    const int plane_cells = n1 * n2;
    const int pt = blockIdx.x, c = blockIdx.z;
    const int jz_begin = blockIdx.y * a.zchunk, jz_end = min(jz_begin + a.zchunk, n0);
    const int f0 = pt * kFwdCells;
    const int chunk0 = (f0 / n2 - 1) * a.d2 - 1;
    const int len = a.len;
    const int dplane = a.d1 * a.d2;
    const int dvol = dplane * a.d0;
    const float* lc = a.lll + (long long)c * dvol;
    const float* hc = a.hf + (long long)c * 7 * dvol;

    auto load_plane = [&](int iz, float (&r)[KI][8]) {    // plane iz of the 8 bands -> registers (zeros outside the level)
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int kk = threadIdx.x + 256 * i;
            const int off = chunk0 + kk;
            const bool ok = kk < len && off >= 0 && off < dplane && iz >= 0 && iz < a.d0;
            const int o = ok ? iz * dplane + off : 0;
            r[i][0] = lc[o];
#pragma unroll
            for (int sb = 1; sb < 8; ++sb) r[i][sb] = hc[(sb - 1) * dvol + o];
            if (DROP) {
                if (a.mul_l) r[i][0] = drop_value(r[i][0], a.mul_l[o], a.thr_l, a.thr_l == a.thr_l);
                if (a.mul_h) {
#pragma unroll
                    for (int sb = 1; sb < 8; ++sb)
                        r[i][sb] = drop_value(r[i][sb], a.mul_h[(sb - 1) * dvol + o], a.thr_h, a.thr_h == a.thr_h);
                }
            }
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) r[i][sb] = ok ? r[i][sb] : 0.0f;
        }
    };
    auto write_plane = [&](int slot, const float (&r)[KI][8]) {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int kk = threadIdx.x + 256 * i;
            if (kk < len) {
                float* rec = s_v + (slot * len + kk) * kRec;
                *reinterpret_cast<f32x4*>(rec) = f32x4{r[i][0], r[i][1], r[i][2], r[i][3]};
                *reinterpret_cast<f32x4*>(rec + 4) = f32x4{r[i][4], r[i][5], r[i][6], r[i][7]};
            }
        }
    };

    const int f = f0 + threadIdx.x;
    const bool in_plane = f < plane_cells;
    const int fc = min(f, plane_cells - 1);
    const int jy = fc / n2, jx = fc - jy * n2;
    const int k00 = jy * a.d2 + jx - chunk0;
    const bool xok0 = jx < a.d2, xok1 = jx >= 1;
    auto read_records = [&](int slot, float (&P)[4][8]) {  // P[q = ey*2+ex][band]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool xok = (q & 1) ? xok1 : xok0;
            const float* rec = s_v + (slot * len + k00 - (q >> 1) * a.d2 - (q & 1)) * kRec;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(rec);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(rec + 4);
            P[q][0] = xok ? lo.x : 0.0f; P[q][1] = xok ? lo.y : 0.0f; P[q][2] = xok ? lo.z : 0.0f; P[q][3] = xok ? lo.w : 0.0f;
            P[q][4] = xok ? hi.x : 0.0f; P[q][5] = xok ? hi.y : 0.0f; P[q][6] = xok ? hi.z : 0.0f; P[q][7] = xok ? hi.w : 0.0f;
        }
    };
    float* outc = a.out + (long long)c * a.t0 * a.t1 * a.t2;
    auto cell = [&](int jz, const float (&E0)[4][8], const float (&E1)[4][8]) {     // E0: plane jz, E1: plane jz - 1
        auto V = [&](int e, int sb) -> float { return (e >> 2) ? E1[e & 3][sb] : E0[e & 3][sb]; };
        float X[2][2][2][2][2];                               // [ez][ey][sz][sy][px]
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ez = q >> 3, ey = (q >> 2) & 1, sz = (q >> 1) & 1, sy = q & 1;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float t = 0.0f;
#pragma unroll
                for (int ex = 0; ex < 2; ++ex)
#pragma unroll
                    for (int sx = 0; sx < 2; ++sx)
                        t = __builtin_fmaf(V(ez * 4 + ey * 2 + ex, sz * 4 + sy * 2 + sx), a.taps[sx * 4 + px + 2 * ex], t);
                X[ez][ey][sz][sy][px] = t;
            }
        }
        float Y[2][2][2][2];                                  // [ez][sz][py][px]
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ez = q >> 3, sz = (q >> 2) & 1, py = (q >> 1) & 1, px = q & 1;
            float t = 0.0f;
#pragma unroll
            for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                for (int sy = 0; sy < 2; ++sy)
                    t = __builtin_fmaf(X[ez][ey][sz][sy][px], a.taps[sy * 4 + py + 2 * ey], t);
            Y[ez][sz][py][px] = t;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pz = p >> 2, py = (p >> 1) & 1, px = p & 1;
            float t = 0.0f;
#pragma unroll
            for (int ez = 0; ez < 2; ++ez)
#pragma unroll
                for (int sz = 0; sz < 2; ++sz)
                    t = __builtin_fmaf(Y[ez][sz][py][px], a.taps[sz * 4 + pz + 2 * ez], t);
            const int oz = 2 * jz + pz - a.o0, oy = 2 * jy + py - a.o1, ox = 2 * jx + px - a.o2;
            if (in_plane && oz >= 0 && oz < a.t0 && oy >= 0 && oy < a.t1 && ox >= 0 && ox < a.t2)
                outc[(oz * a.t1 + oy) * a.t2 + ox] = t;
        }
    };

    float r[KI][8];
    float PA[4][8], PB[4][8];
    load_plane(jz_begin - 1, r);
    write_plane((jz_begin + 1) & 1, r);                   // slot of plane jz_begin - 1
    load_plane(jz_begin, r);
    __syncthreads();
    read_records((jz_begin + 1) & 1, PB);                 // plane jz_begin - 1: the first step's ez = 1 operands
    // two steps per trip so that the register sets alternate roles without copies
#pragma unroll 1
    for (int jz = jz_begin; jz < jz_end; jz += 2) {
        write_plane(jz & 1, r);
        __syncthreads();
        if (jz + 1 < jz_end) load_plane(jz + 1, r);       // in flight under this slice's arithmetic
        read_records(jz & 1, PA);
        cell(jz, PA, PB);
        if (jz + 1 < jz_end) {
            write_plane((jz + 1) & 1, r);
            __syncthreads();
            if (jz + 2 < jz_end) load_plane(jz + 2, r);
            read_records((jz + 1) & 1, PB);
            cell(jz + 1, PB, PA);
        }
    }
}

// Analysis-form kernel shared by the IDWT adjoint and the forward DWT:
//   band_s[c][i] = sum_t src_full[c][2 i + t] * F_s[t],  src_full[u] = src[u - lo] (0 outside [0, n))
// DROP (adjoint with the pruning factors):  stored gradient = band_s * m_s,  d_m_s[i] += band_s[c][i] * coef_s[c][i]
// (float atomics over the channels: whole 256-byte rows per wave instruction, 8..32 adds per address).
// LDS: [512 taps] [6 chunks (source z-plane 2 iz0 - lo0 + zl) of `len` floats]; chunk element k is plane offset
// chunk0 + k, chunk0 = (2 * first cell row - lo1) * n2 - lo2.
struct AnalysisArgs {
    const float* src;      // (C, n0,n1,n2)
    const float* filt;
    float* band0;          // band 0 of channel c at band0 + c * cstride0
    float* bandh;          // band s>=1 of channel c at bandh + c * cstrideh + (s-1) * dvol
    long long cstride0, cstrideh;
    int C, n0, n1, n2, lo0, lo1, lo2, d0, d1, d2;
    int len;
    float taps[8];         // SEP build: [low | high][tap]
    // DROP build only
    const float* lll;      // forward inputs (C, d0,d1,d2), (C, 7, d0,d1,d2): needed for d_mul
    const float* hf;
    const float* mul_l;    // (d0,d1,d2) or NULL
    const float* mul_h;    // (7, d0,d1,d2) or NULL
    float* d_mul_l;        // (d0,d1,d2) or NULL, pre-zeroed
    float* d_mul_h;        // (7, d0,d1,d2) or NULL, pre-zeroed
    // penalty gradients folded in (device scalars or NULL): upstream gradient of  sum coef^2  of the low / detail tensor
    // (adds 2 g coef to its gradient) and of  sum |factor|  of a factor that is itself the penalised parameter
    // (Smallify beta: adds g sign(factor) to the factor gradient, once, by the channel-0 workgroups)
    const float* g_l2_l; const float* g_l2_h; const float* g_l1_l; const float* g_l1_h;
};

__device__ __forceinline__ float sign_of(float v) { return (float)((v > 0.0f) - (v < 0.0f)); }

template <bool DROP, bool SEP>
__global__ __launch_bounds__(256) void analysis_kernel(const AnalysisArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_f = s_dyn;
    float* s_v = s_dyn + 512;
    if (!SEP)
        for (int i = threadIdx.x; i < 512; i += 256) s_f[(i & 63) * 8 + (i >> 6)] = a.filt[i];
    const int plane_cells = a.d1 * a.d2;
    const int pt = blockIdx.x, zt = blockIdx.y, c = blockIdx.z;     // 3-D grid: no index divisions
    const int f0 = pt * kTileCells, iz0 = zt * 2;
    const int chunk0 = (2 * (f0 / a.d2) - a.lo1) * a.n2 - a.lo2;
    const int len = a.len;
    const int nplane = a.n1 * a.n2;
    const long long dvol = (long long)plane_cells * a.d0;
    const float* src = a.src + (long long)c * nplane * a.n0;              // per-channel offsets fit 32 bits (host check)
#pragma unroll 2
    for (int kk = threadIdx.x; kk < len; kk += 256) {
        const int off = chunk0 + kk;
        const bool in_plane = off >= 0 && off < nplane;
        float r[6];
#pragma unroll
        for (int zl = 0; zl < 6; ++zl) {
            const int uz = 2 * iz0 - a.lo0 + zl;
            const bool ok = in_plane && uz >= 0 && uz < a.n0;
            const float x = src[ok ? uz * nplane + off : 0];
            r[zl] = ok ? x : 0.0f;
        }
#pragma unroll
        for (int zl = 0; zl < 6; ++zl) s_v[zl * len + kk] = r[zl];
    }
    __syncthreads();
    const int zl_t = threadIdx.x >> 7;
    const int f = f0 + (threadIdx.x & (kTileCells - 1));
    const int iz = iz0 + zl_t;
    const bool valid = f < plane_cells && iz < a.d0;
    const int fc = min(f, plane_cells - 1);
    const int iy = fc / a.d2, ix = fc - iy * a.d2;
    float acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = 0.0f;
    // every tap of a valid cell lies inside the staged chunk; rows / planes outside the source were staged as zeros,
    // so only the x range needs a select
    const int k0 = (2 * iy - a.lo1) * a.n2 + (2 * ix - a.lo2) - chunk0;
    bool xok[4];
#pragma unroll
    for (int tx = 0; tx < 4; ++tx) { const int ux = 2 * ix + tx - a.lo2; xok[tx] = ux >= 0 && ux < a.n2; }
    if (SEP) {
        // contract x, then y, then z (see idwt_level_kernel): band s = 4 sz + 2 sy + sx
        float Y[4][2][2];                                 // [tz][sy][sx]
#pragma unroll
        for (int tz = 0; tz < 4; ++tz) {
            const float* pl = s_v + (2 * zl_t + tz) * len + k0;
            float X[4][2];                                // [ty][sx]
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                const float* row = pl + ty * a.n2;
                float x0 = 0.0f, x1 = 0.0f;
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) {
                    const float raw = row[tx];                  // always in the chunk: load first, then select (a ternary
                    const float val = xok[tx] ? raw : 0.0f;     // around the load compiles to an exec-masked branch per tap)
                    x0 = __builtin_fmaf(val, a.taps[tx], x0);
                    x1 = __builtin_fmaf(val, a.taps[4 + tx], x1);
                }
                X[ty][0] = x0; X[ty][1] = x1;
            }
#pragma unroll
            for (int sy = 0; sy < 2; ++sy)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    float t = 0.0f;
#pragma unroll
                    for (int ty = 0; ty < 4; ++ty) t = __builtin_fmaf(X[ty][sx], a.taps[sy * 4 + ty], t);
                    Y[tz][sy][sx] = t;
                }
        }
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) {
            float t = 0.0f;
#pragma unroll
            for (int tz = 0; tz < 4; ++tz) t = __builtin_fmaf(Y[tz][(sb >> 1) & 1][sb & 1], a.taps[(sb >> 2) * 4 + tz], t);
            acc[sb] = t;
        }
    } else {
#pragma unroll 1                      // a rolled z-tap loop keeps the live filter taps to one 16-tap plane
    for (int tz = 0; tz < 4; ++tz) {
        const float* pl = s_v + (2 * zl_t + tz) * len + k0;
        float v[16];
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            const float* row = pl + ty * a.n2;
#pragma unroll
            for (int tx = 0; tx < 4; ++tx) { const float raw = row[tx]; v[ty * 4 + tx] = xok[tx] ? raw : 0.0f; }
        }
#pragma unroll
        for (int tyx = 0; tyx < 16; ++tyx) {
            const f32x4 f0v = *reinterpret_cast<const f32x4*>(s_f + (tz * 16 + tyx) * 8);
            const f32x4 f1v = *reinterpret_cast<const f32x4*>(s_f + (tz * 16 + tyx) * 8 + 4);
            acc[0] = __builtin_fmaf(v[tyx], f0v.x, acc[0]); acc[1] = __builtin_fmaf(v[tyx], f0v.y, acc[1]);
            acc[2] = __builtin_fmaf(v[tyx], f0v.z, acc[2]); acc[3] = __builtin_fmaf(v[tyx], f0v.w, acc[3]);
            acc[4] = __builtin_fmaf(v[tyx], f1v.x, acc[4]); acc[5] = __builtin_fmaf(v[tyx], f1v.y, acc[5]);
            acc[6] = __builtin_fmaf(v[tyx], f1v.z, acc[6]); acc[7] = __builtin_fmaf(v[tyx], f1v.w, acc[7]);
        }
    }
    }
    if (valid) {
        const long long sp = (long long)iz * plane_cells + fc;
        if (DROP) {
            if (a.mul_l || a.g_l2_l) {
                const float x = (a.d_mul_l || a.g_l2_l) ? a.lll[(long long)c * dvol + sp] : 0.0f;
                if (a.mul_l) {
                    const float m = a.mul_l[sp];
                    if (a.d_mul_l) atomicAdd(a.d_mul_l + sp, acc[0] * x + ((a.g_l1_l && c == 0) ? *a.g_l1_l * sign_of(m) : 0.0f));
                    acc[0] *= m;
                }
                if (a.g_l2_l) acc[0] = __builtin_fmaf(2.0f * *a.g_l2_l, x, acc[0]);
            }
            if (a.mul_h || a.g_l2_h) {
                const float g2 = a.g_l2_h ? 2.0f * *a.g_l2_h : 0.0f;
                const float g1 = (a.g_l1_h && c == 0) ? *a.g_l1_h : 0.0f;
#pragma unroll
                for (int s = 1; s < 8; ++s) {
                    const long long o = (long long)(s - 1) * dvol + sp;
                    const float x = (a.d_mul_h || a.g_l2_h) ? a.hf[(long long)c * 7 * dvol + o] : 0.0f;
                    if (a.mul_h) {
                        const float m = a.mul_h[o];
                        if (a.d_mul_h) atomicAdd(a.d_mul_h + o, acc[s] * x + g1 * sign_of(m));
                        acc[s] *= m;
                    }
                    acc[s] = __builtin_fmaf(g2, x, acc[s]);
                }
            }
        }
        a.band0[(long long)c * a.cstride0 + sp] = acc[0];
#pragma unroll
        for (int s = 1; s < 8; ++s) a.bandh[(long long)c * a.cstrideh + (long long)(s - 1) * dvol + sp] = acc[s];
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

inline int check_level(const void* a, const void* b, const void* c, const void* d, int C, int d0, int d1, int d2,
                       int t0, int t1, int t2) {
    if (!a || !b || !c || !d) return LFGC_E_NULL;
    if (C < 1 || d0 < 1 || d1 < 1 || d2 < 1 || t0 < 1 || t1 < 1 || t2 < 1) return LFGC_E_SHAPE;
    if (t0 > 2 * d0 + 2 || t1 > 2 * d1 + 2 || t2 > 2 * d2 + 2) return LFGC_E_SHAPE;
    return LFGC_OK;
}

template <typename K, typename A>
int launch_tiled(K kern, int* lds_limit, const A& a, dim3 blocks, int lds_bytes, hipStream_t stream) {
    if (blocks.y > 65535u || blocks.z > 65535u || lds_bytes > 160 * 1024) return LFGC_E_UNSUPPORTED;
    if (lds_bytes > 64 * 1024 && lds_bytes > *lds_limit) {      // raised once per kernel: launches stay graph-capturable
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        *lds_limit = lds_bytes;
    }
    hipLaunchKernelGGL(kern, blocks, dim3(256), lds_bytes, stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

int launch_idwt(IdwtArgs a, bool drop, const float* taps, hipStream_t stream) {
    if (taps) for (int i = 0; i < 8; ++i) a.taps[i] = taps[i];
    a.o0 = (2 * a.d0 + 2 - a.t0) / 2; a.o1 = (2 * a.d1 + 2 - a.t1) / 2; a.o2 = (2 * a.d2 + 2 - a.t2) / 2;
    const int n0 = a.d0 + 1, n1 = a.d1 + 1, n2 = a.d2 + 1;
    const int span = (kFwdCells + n2 - 2) / n2;             // rows a run of 256 cells can straddle beyond its first
    a.len = (span + 2) * a.d2 + 2;
    const long long ptiles = ((long long)n1 * n2 + kFwdCells - 1) / kFwdCells;
    if (ptiles > 0x7fffffffLL) return LFGC_E_UNSUPPORTED;
    const dim3 blocks((unsigned)ptiles, (unsigned)((n0 + 1) / 2), (unsigned)a.C);
    const int lds = (512 + 3 * a.len * kRec) * 4;
    if ((long long)a.d0 * a.d1 * a.d2 > 0x7fffffffLL / 8 || (long long)a.t0 * a.t1 * a.t2 > 0x7fffffffLL) return LFGC_E_UNSUPPORTED;
    static int lim[4][LFGC_MAX_DEVICES] = {{0}};       // per (kernel, device)
    const int dev_lim = lfgc_current_device();
    const int ki = (a.len + 255) / 256;
    // Small levels (<= 40 000 output voxels per channel: everything below the last two levels of a 64^3 grid) take the tiled
    // kernel below: one load -> barrier -> stencil -> store round per workgroup instead of a z walk whose every step waits
    // for a plane (the arithmetic that should hide it is nothing at these sizes): cfg-3 train step 0.385 -> 0.378 ms.
    const bool small_level = (long long)a.t0 * a.t1 * a.t2 <= 40000;
    if (taps && ki <= 3 && !small_level) {                   // separable filter: sliding window along z
        a.zchunk = n0 < 6 ? n0 : 5;
        const dim3 sblocks((unsigned)ptiles, (unsigned)((n0 + a.zchunk - 1) / a.zchunk), (unsigned)a.C);
        const int slds = 2 * a.len * kRec * 4;
        static int slim[6][LFGC_MAX_DEVICES] = {{0}};       // per (kernel, device)
    const int dev_slim = lfgc_current_device();
        if (drop) {
            if (ki == 1) return launch_tiled(idwt_slide_kernel<true, 1>, &slim[0][dev_slim], a, sblocks, slds, stream);
            if (ki == 2) return launch_tiled(idwt_slide_kernel<true, 2>, &slim[1][dev_slim], a, sblocks, slds, stream);
            return launch_tiled(idwt_slide_kernel<true, 3>, &slim[2][dev_slim], a, sblocks, slds, stream);
        }
        if (ki == 1) return launch_tiled(idwt_slide_kernel<false, 1>, &slim[3][dev_slim], a, sblocks, slds, stream);
        if (ki == 2) return launch_tiled(idwt_slide_kernel<false, 2>, &slim[4][dev_slim], a, sblocks, slds, stream);
        return launch_tiled(idwt_slide_kernel<false, 3>, &slim[5][dev_slim], a, sblocks, slds, stream);
    }
    if (taps) return drop ? launch_tiled(idwt_level_kernel<true, true>, &lim[3][dev_lim], a, blocks, lds, stream)
                          : launch_tiled(idwt_level_kernel<false, true>, &lim[2][dev_lim], a, blocks, lds, stream);
    return drop ? launch_tiled(idwt_level_kernel<true, false>, &lim[1][dev_lim], a, blocks, lds, stream)
                : launch_tiled(idwt_level_kernel<false, false>, &lim[0][dev_lim], a, blocks, lds, stream);
}

int launch_analysis(AnalysisArgs a, bool drop, const float* taps, hipStream_t stream) {
    if (taps) for (int i = 0; i < 8; ++i) a.taps[i] = taps[i];
    const int span = (kTileCells + a.d2 - 2) / a.d2;
    a.len = (2 * span + 3) * a.n2 + 2 * a.d2 + 2;
    const long long ptiles = ((long long)a.d1 * a.d2 + kTileCells - 1) / kTileCells;
    if (ptiles > 0x7fffffffLL) return LFGC_E_UNSUPPORTED;
    const dim3 blocks((unsigned)ptiles, (unsigned)((a.d0 + 1) / 2), (unsigned)a.C);
    const int lds = (512 + 6 * a.len) * 4;
    if ((long long)a.n0 * a.n1 * a.n2 > 0x7fffffffLL) return LFGC_E_UNSUPPORTED;
    static int lim[4][LFGC_MAX_DEVICES] = {{0}};       // per (kernel, device)
    const int dev_lim = lfgc_current_device();
    if (taps) return drop ? launch_tiled(analysis_kernel<true, true>, &lim[3][dev_lim], a, blocks, lds, stream)
                          : launch_tiled(analysis_kernel<false, true>, &lim[2][dev_lim], a, blocks, lds, stream);
    return drop ? launch_tiled(analysis_kernel<true, false>, &lim[1][dev_lim], a, blocks, lds, stream)
                : launch_tiled(analysis_kernel<false, false>, &lim[0][dev_lim], a, blocks, lds, stream);
}

}  // namespace

extern "C" int lfgc_idwt_level_f32(const float* lll, const float* hf, const float* filter_rev, const float* taps, float* out,
                                   int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(lll, hf, taps ? (const void*)taps : (const void*)filter_rev, out, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    IdwtArgs a = {};
    a.lll = lll; a.hf = hf; a.filt = filter_rev; a.out = out;
    a.C = C; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
    return launch_idwt(a, false, taps, (hipStream_t)stream);
}

extern "C" int lfgc_idwt_level_drop_f32(const float* lll, const float* hf, const float* mul_lll, float thr_lll,
                                        const float* mul_hf, float thr_hf, const float* filter_rev, const float* taps, float* out,
                                        int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(lll, hf, taps ? (const void*)taps : (const void*)filter_rev, out, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    IdwtArgs a = {};
    a.lll = lll; a.hf = hf; a.filt = filter_rev; a.out = out;
    a.C = C; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
    a.mul_l = mul_lll; a.mul_h = mul_hf; a.thr_l = thr_lll; a.thr_h = thr_hf;
    return launch_idwt(a, mul_lll || mul_hf, taps, (hipStream_t)stream);
}

static AnalysisArgs adjoint_args(const float* d_out, const float* filter_rev, float* d_lll, float* d_hf,
                                 int C, int d0, int d1, int d2, int t0, int t1, int t2) {
    AnalysisArgs a = {};
    a.src = d_out; a.filt = filter_rev; a.band0 = d_lll; a.bandh = d_hf;
    const long long dvol = (long long)d0 * d1 * d2;
    a.cstride0 = dvol; a.cstrideh = 7 * dvol;
    a.C = C; a.n0 = t0; a.n1 = t1; a.n2 = t2;
    a.lo0 = (2 * d0 + 2 - t0) / 2; a.lo1 = (2 * d1 + 2 - t1) / 2; a.lo2 = (2 * d2 + 2 - t2) / 2;
    a.d0 = d0; a.d1 = d1; a.d2 = d2;
    return a;
}

extern "C" int lfgc_idwt_level_bwd_f32(const float* d_out, const float* filter_rev, const float* taps, float* d_lll, float* d_hf,
                                       int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(d_out, taps ? (const void*)taps : (const void*)filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    return launch_analysis(adjoint_args(d_out, filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2), false, taps, (hipStream_t)stream);
}

extern "C" int lfgc_idwt_level_drop_bwd_f32(const float* d_out, const float* filter_rev, const float* taps, const float* lll, const float* hf,
                                            const float* mul_lll, const float* mul_hf, float* d_lll, float* d_hf,
                                            float* d_mul_lll, float* d_mul_hf, const float* const* penalty_grads,
                                            int C, int d0, int d1, int d2,
                                            int t0, int t1, int t2, lfgc_stream_t stream) {
    const int rc = check_level(d_out, taps ? (const void*)taps : (const void*)filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    if ((d_mul_lll && (!mul_lll || !lll)) || (d_mul_hf && (!mul_hf || !hf))) return LFGC_E_NULL;
    const float* pg[4] = {nullptr, nullptr, nullptr, nullptr};
    if (penalty_grads) for (int i = 0; i < 4; ++i) pg[i] = penalty_grads[i];
    if ((pg[0] && !lll) || (pg[1] && !hf) || (pg[2] && !d_mul_lll) || (pg[3] && !d_mul_hf)) return LFGC_E_NULL;
    AnalysisArgs a = adjoint_args(d_out, filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2);
    a.lll = lll; a.hf = hf; a.mul_l = mul_lll; a.mul_h = mul_hf; a.d_mul_l = d_mul_lll; a.d_mul_h = d_mul_hf;
    a.g_l2_l = pg[0]; a.g_l2_h = pg[1]; a.g_l1_l = pg[2]; a.g_l1_h = pg[3];
    return launch_analysis(a, mul_lll || mul_hf || pg[0] || pg[1], taps, (hipStream_t)stream);
}

extern "C" int lfgc_dwt_level_f32(const float* in, const float* filter_fwd, const float* taps, float* out,
                                  int C, int n0, int n1, int n2, lfgc_stream_t stream) {
    if (!in || (!filter_fwd && !taps) || !out) return LFGC_E_NULL;
    if (C < 1 || n0 < 1 || n1 < 1 || n2 < 1) return LFGC_E_SHAPE;
    // _get_padding_size (Torch_Wavelet_Transform.py:59-63): F.pad slots are (last axis lo, hi, ..., first axis
    // lo, hi) while is_odd is indexed first axis first -> the odd bit of axis a pads axis 2-a.
    const int hi0 = 2 + (n2 & 1), hi1 = 2 + (n1 & 1), hi2 = 2 + (n0 & 1);
    AnalysisArgs a = {};
    a.src = in; a.filt = filter_fwd;
    a.d0 = (n0 + 2 + hi0 - 4) / 2 + 1; a.d1 = (n1 + 2 + hi1 - 4) / 2 + 1; a.d2 = (n2 + 2 + hi2 - 4) / 2 + 1;
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    a.band0 = out; a.bandh = out + dvol;
    a.cstride0 = 8 * dvol; a.cstrideh = 8 * dvol;
    a.C = C; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.lo0 = 2; a.lo1 = 2; a.lo2 = 2;
    return launch_analysis(a, false, taps, (hipStream_t)stream);
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
