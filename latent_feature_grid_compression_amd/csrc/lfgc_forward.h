// lfgc_forward.h -- fused trilinear sample + Fourier embed + SnakeAlt MLP forward for gfx950.
//
// Replaces model/Feature_Grid_Model.py:62-78 of the reference (grid_sample, Embedder.embed, cat,
// L x (Linear + SnakeAlt), final Linear, optional clamp).
//
// Mapping (one wave = 32 samples, v_mfma_f32_32x32x2_f32, exact fp32):
//   Out^T[h_out, sample] = W[h_out, k] . In^T[k, sample]
//   A operand = W tile from LDS      lane (i = lane&31, hh = lane>>5) supplies W[32m + i][k(s, hh)]
//   B operand = activations in VGPRs lane (j = lane&31, hh)           supplies In[k(s, hh)][sample j]
//   C/D       = 16 accumulators      lane (j, hh), reg r = row 32m + (r&3) + 8(r>>2) + 4hh of sample j
// The k order is chosen as k(s, hh) = 32(s>>4) + 8((s>>2)&3) + 4hh + (s&3), which is exactly the row a
// lane's accumulator register r = s&15 of tile m = s>>4 holds: after SnakeAlt the accumulators ARE the
// next layer's B operands, so activations never leave registers and never cross lanes.  With that
// order a lane's A values for 4 consecutive k-steps are 4 consecutive floats of one W row: one
// ds_read_b128 feeds 4 MFMAs.
// Nets whose layer blocks all fit 80 KB of LDS: workgroup = 4 waves = 128 samples, blocks staged once, two
// workgroups per CU.  Larger nets: workgroup = 8 waves = 256 samples, one per CU, layer blocks stream through a
// 2-deep LDS ring by LDS-DMA (next layer in flight while the current one computes, one barrier per layer).
#pragma once
#include "lfgc_common.h"

// Diagnostics only (tools/ablate_forward.py builds separate libraries with -DLFGC_ABLATE=mask; the shipped
// library is always built with 0): 1 = no gather loads, 2 = activation replaced by 0.5*a, 4 = no MFMAs.
#ifndef LFGC_ABLATE
#define LFGC_ABLATE 0
#endif

struct LfgcFwdArgs {
    const float* pos;          // (N,3) or nullptr (lattice mode)
    long long n;               // samples
    int res0, res1, res2;      // lattice mode: volume resolution
    int x_begin;               // lattice mode: first x of the slab
    int tile;                  // lattice mode: tile edge (32)
    float scale0, scale1, scale2;   // lattice mode: dataset.scales (data/IndexDataset.py:64-65)
    const float* grid;         // (D,H,W,Cs)
    int D, H, W, Cs;
    const float* packed;
    int L;
    int resident;              // 1: all layer blocks staged once
    int clamp;
    float* out;                // (N)
    float* stash;              // or nullptr
    long long nbatches;        // ceil(N / (32 * waves per workgroup))
    int waves;                 // waves per workgroup of the chosen build (4 or 8)
    int coord_table;           // lattice mode: per-axis coordinate tables fit LDS (res0+res1+res2 floats)
    int single;                // f16 builds: 1 = single product W_hi.h_hi (LFGC_PRECISION_F16), 0 = hi/lo split
    int* status;               // f16 builds: set to 1 when a sample left the range of the fast arithmetic (or nullptr)
    const int* redo_if;        // exact build: run only if *redo_if != 0 (nullptr: always) -- the range fallback
    unsigned long long* stamps;   // diagnostics builds (-DLFGC_STAMPS, tools/phase_stamps.py): per-wave cycle totals per phase
    // lattice mode, f16 builds: "z-run" tiles (LfgcColumnSampler): a wave's 32 samples are 32 consecutive z of ONE (x, y)
    // row of the slab; tiles_per_row = ceil(res2 / 32), ntiles = rows * tiles_per_row, nzc = z cells a tile's column holds
    int zrun, nzc, tiles_per_row;
    long long ntiles;
    int x2;                    // z-run launches: the two-tiles-per-wave kernel (lfgc_forward16x2.h); nbatches counts passes of 8 tiles
};

// Lattice coordinate of voxel v along one axis, formed like field_from_net does per tile
// (visualization/OutputToVTK.py:23-37) + torch.linspace's CPU formula (start + step*i below the
// midpoint, end - step*(n-1-i) above).
__device__ __forceinline__ float lfgc_lattice_coord(int v, int res, int tile, float scale) {
    const int tb = (v / tile) * tile;
    const int te = min(tb + tile, res);
    const int cnt = te - tb;
    const int i = v - tb;
    const float max_idx = (float)(res - 1);
    const float min_alpha = (float)((double)tb / (double)(res - 1));
    const float max_alpha = (float)((double)(te - 1) / (double)(res - 1));
    const float min_bounds = __fadd_rn(0.0f, __fmul_rn(min_alpha, max_idx));
    const float max_bounds = __fadd_rn(0.0f, __fmul_rn(max_alpha, max_idx));
    const float start = __fdiv_rn(min_bounds, max_idx);
    const float end = __fdiv_rn(max_bounds, max_idx);
    float lin;
    if (cnt == 1) {
        lin = start;
    } else {
        const float step = __fdiv_rn(__fsub_rn(end, start), (float)(cnt - 1));
        lin = (i < cnt / 2) ? __fadd_rn(start, __fmul_rn(step, (float)i))
                            : __fsub_rn(end, __fmul_rn(step, (float)(cnt - i - 1)));
    }
    const float nrm = __fsub_rn(__fmul_rn(2.0f, lin), 1.0f);
    return __fmul_rn(scale, nrm);
}

// Scalar inputs [p | sin f_k p | cos f_k p | 0 pad] of the lane's sample: a lane keeps only its half of the list
// (E[0 .. EP/2) = entries [hh EP/2, (hh+1) EP/2)).
template <int NF>
__device__ __forceinline__ void lfgc_embed_inputs(float p0, float p1, float p2, int hh, float* __restrict__ E) {
    constexpr int EE = 3 + 6 * NF;
    constexpr int EP = (EE + 7) / 8 * 8;
    constexpr int EPH = EP / 2;
    constexpr int E_ = EE;

        // ---- scalar inputs [p | sin f_k p | cos f_k p]: a lane keeps only its half of the list ---------------------
        if constexpr (NF == 2) {
            // e = [p0 p1 p2 s0x s0y s0z c0x c0y | c0z s1x s1y s1z c1x c1y c1z 0] (f0, f1 = 2 f0): lane half 0 needs sin/cos
            // of f0 p, lane half 1 those of f1 p plus cos(f0 p2) -- three sincos of lane-half-dependent arguments and one
            // cosine instead of six sincos per lane.  Arguments are formed as before (one fp32 product each).
            const float f0 = lfgc_freq(0), f1 = lfgc_freq(1);
            const float f = hh ? f1 : f0;
            const float a0 = __fmul_rn(p0, f), a1 = __fmul_rn(p1, f), a2 = __fmul_rn(p2, f), a3 = __fmul_rn(p2, f0);
            const bool bad = lfgc_trig_out_of_range(a0) | lfgc_trig_out_of_range(a1) | lfgc_trig_out_of_range(a2) |
                             lfgc_trig_out_of_range(a3);
            float sA, cA, sB, cB, sC, cC, sD, cD;
            lfgc_sincosf_t<false>(a0, sA, cA);
            lfgc_sincosf_t<false>(a1, sB, cB);
            lfgc_sincosf_t<false>(a2, sC, cC);
            cD = lfgc_cosf_t<false>(a3);
            if (__builtin_expect(__any(bad), 0)) {   // positions far outside [-1,1], inf or nan
                lfgc_sincosf_t<true>(a0, sA, cA);
                lfgc_sincosf_t<true>(a1, sB, cB);
                lfgc_sincosf_t<true>(a2, sC, cC);
                lfgc_sincosf_t<true>(a3, sD, cD);
            }
            const float lo[8] = {p0, p1, p2, sA, sB, sC, cA, cB};
            const float hi[8] = {cD, sA, sB, sC, cA, cB, cC, 0.0f};
#pragma unroll
            for (int t = 0; t < EPH; ++t) E[t] = hh ? hi[t] : lo[t];
        } else {
            float e[EP];
            e[0] = p0; e[1] = p1; e[2] = p2;
            bool bad = false;
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const float f = lfgc_freq(k);
                const float a0 = __fmul_rn(p0, f), a1 = __fmul_rn(p1, f), a2 = __fmul_rn(p2, f);
                bad |= lfgc_trig_out_of_range(a0) | lfgc_trig_out_of_range(a1) | lfgc_trig_out_of_range(a2);
                float s, c;
                lfgc_sincosf_t<false>(a0, s, c); e[3 + 6 * k + 0] = s; e[3 + 6 * k + 3] = c;
                lfgc_sincosf_t<false>(a1, s, c); e[3 + 6 * k + 1] = s; e[3 + 6 * k + 4] = c;
                lfgc_sincosf_t<false>(a2, s, c); e[3 + 6 * k + 2] = s; e[3 + 6 * k + 5] = c;
            }
            if (__builtin_expect(__any(bad), 0)) {   // positions far outside [-1,1], inf or nan
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const float f = lfgc_freq(k);
                    float s, c;
                    lfgc_sincosf_t<true>(__fmul_rn(p0, f), s, c); e[3 + 6 * k + 0] = s; e[3 + 6 * k + 3] = c;
                    lfgc_sincosf_t<true>(__fmul_rn(p1, f), s, c); e[3 + 6 * k + 1] = s; e[3 + 6 * k + 4] = c;
                    lfgc_sincosf_t<true>(__fmul_rn(p2, f), s, c); e[3 + 6 * k + 2] = s; e[3 + 6 * k + 5] = c;
                }
            }
#pragma unroll
            for (int t = E_; t < EP; ++t) e[t] = 0.0f;
#pragma unroll
            for (int t = 0; t < EPH; ++t) {
                float lo = e[t], hi = e[EPH + t];
                asm volatile("" : "+v"(lo), "+v"(hi));     // keep both in VGPRs: a select of two array slots would go to scratch
                E[t] = hh ? hi : lo;
            }
        }
}

// Positions, trilinear gather and Fourier embedding for the lane's sample, in two phases so that the gather's 8 x CH/8
// 16-byte loads are in flight while the caller does something else (the f16 build puts its layer-0 barrier there):
//   issue():  position, ATen's unnormalise / floor / corner weights (zero padding = zero weight), the 8 corner rows
//             requested into registers;
//   finish(): X[0..CH/2) = interpolated grid channels [hh*CH/2, (hh+1)*CH/2), X[CH/2 .. CH/2 + EP/2) = this lane half's
//             share of [p | sin f_k p | cos f_k p | 0 pad].
// Corner offsets are 32-bit (the host entry refuses grids of 4 GiB or more); all conditions are evaluated as masks,
// not short-circuit branches.
template <int CH, int NF>
struct LfgcSampler {
    static constexpr int E = 3 + 6 * NF;
    static constexpr int EP = (E + 7) / 8 * 8;
    static constexpr int CHH = CH / 2;
    static constexpr int EPH = EP / 2;
    float p0, p1, p2;
    float w[8];
    f32x4 v[8][CHH / 4];

    __device__ __forceinline__ void issue(const LfgcFwdArgs& a, long long nc, long long N, const float* s_coord, int hh) {
        // ---- positions ---------------------------------------------------------------------------
        if (a.pos) {
            const float* pp = a.pos + 3 * nc;
            p0 = pp[0]; p1 = pp[1]; p2 = pp[2];
        } else {
            int vx, vy, vz;
            if (N <= 0xffffffffLL) {                          // 32-bit index arithmetic (any slab up to 1625^3)
                const unsigned plane = (unsigned)a.res1 * (unsigned)a.res2;
                const unsigned un = (unsigned)nc;
                const unsigned qx = un / plane, rem = un - qx * plane;
                const unsigned qy = rem / (unsigned)a.res2;
                vx = a.x_begin + (int)qx; vy = (int)qy; vz = (int)(rem - qy * (unsigned)a.res2);
            } else {
                const long long plane = (long long)a.res1 * a.res2;
                vx = a.x_begin + (int)(nc / plane);
                const long long rem = nc % plane;
                vy = (int)(rem / a.res2); vz = (int)(rem % a.res2);
            }
            if (a.coord_table) {
                p0 = s_coord[vx]; p1 = s_coord[a.res0 + vy]; p2 = s_coord[a.res0 + a.res1 + vz];
            } else {
                p0 = lfgc_lattice_coord(vx, a.res0, a.tile, a.scale0);
                p1 = lfgc_lattice_coord(vy, a.res1, a.tile, a.scale1);
                p2 = lfgc_lattice_coord(vz, a.res2, a.tile, a.scale2);
            }
        }
        // ---- trilinear gather: lane (j, hh) interpolates channels [hh*CHH, (hh+1)*CHH) -------------
        // grid_sampler_unnormalize, align_corners=False: ((p + 1) * size - 1) / 2   (ATen GridSampler.h)
        const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p0, 1.0f), (float)a.W), 1.0f), 0.5f);
        const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p1, 1.0f), (float)a.H), 1.0f), 0.5f);
        const float iz = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p2, 1.0f), (float)a.D), 1.0f), 0.5f);
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        // clamped to [-2, size] before the conversion: absurd / NaN positions stay defined, and every corner of such a
        // sample is then out of bounds (weight 0) by the index test alone
        const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)a.W);
        const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)a.H);
        const int z0 = (int)fminf(fmaxf(fz0, -2.0f), (float)a.D);
        float wx[2], wy[2], wz[2];
        wx[1] = __fsub_rn(ix, fx0); wx[0] = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
        wy[1] = __fsub_rn(iy, fy0); wy[0] = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
        wz[1] = __fsub_rn(iz, fz0); wz[0] = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
        unsigned ox[2], oy[2], oz[2];                   // BYTE offsets (32-bit: one SGPR base + one VGPR offset per load)
        const unsigned cell = 4u * (unsigned)a.Cs, row = (unsigned)a.W * cell, plane = (unsigned)a.H * row;
#pragma unroll
        for (int d = 0; d < 2; ++d) {          // out-of-bounds corner = zero weight on that axis; its (clamped) row is still read
            wx[d] = ((unsigned)(x0 + d) < (unsigned)a.W) ? wx[d] : 0.0f;
            wy[d] = ((unsigned)(y0 + d) < (unsigned)a.H) ? wy[d] : 0.0f;
            wz[d] = ((unsigned)(z0 + d) < (unsigned)a.D) ? wz[d] : 0.0f;
            ox[d] = (unsigned)min(max(x0 + d, 0), a.W - 1) * cell + (unsigned)(hh * CHH * 4);
            oy[d] = (unsigned)min(max(y0 + d, 0), a.H - 1) * row;
            oz[d] = (unsigned)min(max(z0 + d, 0), a.D - 1) * plane;
        }
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int dz = corner >> 2, dy = (corner >> 1) & 1, dx = corner & 1;   // ATen order: tnw, tne, tsw, tse, bnw, ...
            w[corner] = __fmul_rn(__fmul_rn(wx[dx], wy[dy]), wz[dz]);
#if !(LFGC_ABLATE & 1)
            const float* gp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.grid) + (oz[dz] + oy[dy] + ox[dx]));
#pragma unroll
            for (int c4 = 0; c4 < CHH / 4; ++c4) v[corner][c4] = *reinterpret_cast<const f32x4*>(gp + 4 * c4);
#else
#pragma unroll
            for (int c4 = 0; c4 < CHH / 4; ++c4) v[corner][c4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#endif
        }
    }

    __device__ __forceinline__ void finish(int hh, float (&B0)[CHH + EPH]) {
        lfgc_embed_inputs<NF>(p0, p1, p2, hh, B0 + CHH);
        // ---- the 8 corner rows, accumulated in ATen's corner order -----------------------------------
        float feat[CHH];
#pragma unroll
        for (int c = 0; c < CHH; ++c) feat[c] = 0.0f;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
#pragma unroll
            for (int c4 = 0; c4 < CHH / 4; ++c4) {
                const f32x4 q = v[corner][c4];
                feat[4 * c4 + 0] = __builtin_fmaf(q.x, w[corner], feat[4 * c4 + 0]);
                feat[4 * c4 + 1] = __builtin_fmaf(q.y, w[corner], feat[4 * c4 + 1]);
                feat[4 * c4 + 2] = __builtin_fmaf(q.z, w[corner], feat[4 * c4 + 2]);
                feat[4 * c4 + 3] = __builtin_fmaf(q.w, w[corner], feat[4 * c4 + 3]);
            }
        }
#pragma unroll
        for (int c = 0; c < CHH; ++c) B0[c] = feat[c];
    }
};

// Lattice mode, "z-run" tiles: the 32 samples of a wave are 32 consecutive z voxels of ONE (x, y) row of the volume, so
// they share their x and y cells and weights and together touch only a short COLUMN of the grid: the cells
// zc_lo .. zc_lo + nzc - 1 along z at the four (x, y) corners (nzc = floor(31 D / (res2 - 1)) + 3: 10 cells of the 64-cell
// axis for a 256-voxel row, against 32 samples x 8 corner rows read one by one by LfgcSampler).  The wave
//   stage_a: forms the positions and ATen's weights as LfgcSampler does, then -- 64 lanes = cells x 4-channel quads --
//            loads the four (x, y) corner rows of every cell of the column (4 x dwordx4 per lane and pass instead of 32),
//            contracts them with the x-y weights (ATen's corner order nw, ne, sw, se; the z weight is applied afterwards,
//            once per plane: a different but equally short rounding sequence) and leaves the column in LDS;
//   stage_b: every lane reads the two planes of ITS sample back (2 x CH/8 ds_read_b128) and applies the z weights.
// 16 + 32 FMAs per lane instead of 128, 4-8 vector loads per lane instead of 32, and the loads go out once per cell
// instead of once per sample: the texture-address path, which the 8 waves of a workgroup used to fill for 4 k cycles per
// batch (32 x 8 x 1 KiB through 64 B / clock), is nearly idle.  Lanes whose cell index lies past the column redo one of
// its cells (same value to the same LDS address): no lane predicate, no branch.
template <int CH, int NF>
struct LfgcColumnSampler {
    static constexpr int E = 3 + 6 * NF;
    static constexpr int EP = (E + 7) / 8 * 8;
    static constexpr int CHH = CH / 2;
    static constexpr int EPH = EP / 2;
    static constexpr int LPC = CH / 4;              // lanes per cell: one 4-channel quad each
    static constexpr int CPP = 64 / LPC;            // cells per pass of the wave
    static constexpr int NZC_MAX = 12;              // longest column the host selects this path for
    static constexpr int NPASS = (NZC_MAX + CPP - 1) / CPP;
    // LDS row of one column cell: CH floats + 4 of padding -- with 128-byte rows, rows r and r + 2 start on the same bank
    // and the read-back (16 lanes of a ds_read_b128 group spread over 4-5 neighbouring rows) conflicted 2-3 ways
    static constexpr int CS = CH + 4;
    float p0, p1, p2;
    float wz0, wz1;
    int zrel;                                       // (z cell of the sample) - zc_lo, in [0, nzc - 2]
    int q, kcell[NPASS];                            // the lane's channel quad and column cell per pass
    float wxy_[4];
    f32x4 v[NPASS][4];                              // corner quads in flight between stage_a and stage_b

    // vx, vy: the row (wave-uniform); vz: the lane's voxel, already clamped to the row.  s_col: this wave's nzc x CH floats.
    __device__ __forceinline__ void stage_a(const LfgcFwdArgs& a, int vx, int vy, int vz, const float* s_coord,
                                            float* s_col, int lane) {
        p0 = s_coord[vx]; p1 = s_coord[a.res0 + vy]; p2 = s_coord[a.res0 + a.res1 + vz];
        // grid_sampler_unnormalize, align_corners=False: ((p + 1) * size - 1) / 2   (ATen GridSampler.h)
        const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p0, 1.0f), (float)a.W), 1.0f), 0.5f);
        const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p1, 1.0f), (float)a.H), 1.0f), 0.5f);
        const float iz = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p2, 1.0f), (float)a.D), 1.0f), 0.5f);
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)a.W);
        const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)a.H);
        const int z0 = (int)fminf(fmaxf(fz0, -2.0f), (float)a.D);
        float wx[2], wy[2];
        wx[1] = __fsub_rn(ix, fx0); wx[0] = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
        wy[1] = __fsub_rn(iy, fy0); wy[0] = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
        wz1 = __fsub_rn(iz, fz0); wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
        unsigned ox[2], oy[2];
        const unsigned cell = 4u * (unsigned)a.Cs, row = (unsigned)a.W * cell, plane = (unsigned)a.H * row;
#pragma unroll
        for (int d = 0; d < 2; ++d) {          // out-of-bounds corner = zero weight on that axis; its (clamped) row is still read
            wx[d] = ((unsigned)(x0 + d) < (unsigned)a.W) ? wx[d] : 0.0f;
            wy[d] = ((unsigned)(y0 + d) < (unsigned)a.H) ? wy[d] : 0.0f;
            ox[d] = (unsigned)min(max(x0 + d, 0), a.W - 1) * cell;
            oy[d] = (unsigned)min(max(y0 + d, 0), a.H - 1) * row;
        }
        wz0 = ((unsigned)z0 < (unsigned)a.D) ? wz0 : 0.0f;
        wz1 = ((unsigned)(z0 + 1) < (unsigned)a.D) ? wz1 : 0.0f;
        const int zc_lo = __builtin_amdgcn_readfirstlane(z0);          // lane 0 holds the row's smallest z
        zrel = min(max(z0 - zc_lo, 0), a.nzc - 2);
        const float wxy[4] = {__fmul_rn(wx[0], wy[0]), __fmul_rn(wx[1], wy[0]), __fmul_rn(wx[0], wy[1]), __fmul_rn(wx[1], wy[1])};
        const unsigned oxy[4] = {oy[0] + ox[0], oy[0] + ox[1], oy[1] + ox[0], oy[1] + ox[1]};
        q = lane % LPC;
        const int kl = lane / LPC;
        const float inv_nzc = 1.0f / (float)a.nzc;
#pragma unroll
        for (int c = 0; c < 4; ++c) wxy_[c] = wxy[c];
        // every pass's corner quads are requested here and contracted in stage_b: their latency runs under the embedding
        // arithmetic and the layer-0 barrier instead of being waited out pass by pass
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            int k = ps * CPP + kl;                                     // < NPASS * CPP + 64 / LPC <= 64: exact in fp32
            k -= a.nzc * (int)(((float)k + 0.5f) * inv_nzc);           // k mod nzc
            kcell[ps] = k;
            const int zc = min(max(zc_lo + k, 0), a.D - 1);
            const unsigned off = (unsigned)zc * plane + (unsigned)(q * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[ps][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.grid) + (off + oxy[c]));
        }
    }

    __device__ __forceinline__ void stage_b(float* s_col, int hh, float (&B0)[CHH + EPH]) {
        lfgc_embed_inputs<NF>(p0, p1, p2, hh, B0 + CHH);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc.x = __builtin_fmaf(v[ps][c].x, wxy_[c], acc.x); acc.y = __builtin_fmaf(v[ps][c].y, wxy_[c], acc.y);
                acc.z = __builtin_fmaf(v[ps][c].z, wxy_[c], acc.z); acc.w = __builtin_fmaf(v[ps][c].w, wxy_[c], acc.w);
            }
            *reinterpret_cast<f32x4*>(s_col + kcell[ps] * CS + 4 * q) = acc;
        }
        const float* c0 = s_col + zrel * CS + hh * CHH;
#pragma unroll
        for (int c4 = 0; c4 < CHH / 4; ++c4) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(c0 + 4 * c4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(c0 + CS + 4 * c4);
            B0[4 * c4 + 0] = __builtin_fmaf(hi.x, wz1, __fmul_rn(lo.x, wz0));
            B0[4 * c4 + 1] = __builtin_fmaf(hi.y, wz1, __fmul_rn(lo.y, wz0));
            B0[4 * c4 + 2] = __builtin_fmaf(hi.z, wz1, __fmul_rn(lo.z, wz0));
            B0[4 * c4 + 3] = __builtin_fmaf(hi.w, wz1, __fmul_rn(lo.w, wz0));
        }
    }
};

template <int CH, int NF>
__device__ __forceinline__ void lfgc_sample_inputs(const LfgcFwdArgs& a, long long nc, long long N, const float* s_coord,
                                               int hh, float (&B0)[CH / 2 + ((3 + 6 * NF + 7) / 8 * 8) / 2]) {
    LfgcSampler<CH, NF> sm;
    sm.issue(a, nc, N, s_coord, hh);
    sm.finish(hh, B0);
}

// One hidden layer on a 32-sample tile held in registers.
//   s_blk : LDS block [W (32*MT rows x S floats) | bias 32*MT]
//   Bin   : KS activations of this lane (k order above);  Bout : 16*MT outputs (same order)
//   stash : this layer's slot for the tile: [(m*16 + r)][64 lanes]  (STASH builds only)
template <int KS, int MT, int S, bool STASH>
__device__ __forceinline__ void lfgc_layer_fwd(const float* __restrict__ s_blk, const float (&Bin)[KS],
                                               float (&Bout)[16 * MT], float* __restrict__ stash,
                                               int j, int hh, int lane) {
    static_assert(KS % 4 == 0, "k-steps come in groups of 4 (one ds_read_b128)");
    const float* s_bias = s_blk + 32 * MT * S + 4 * hh;
    const float* s_row = s_blk + j * S + 4 * hh;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + 32 * m + 8 * q);
            acc[4 * q + 0] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
        }
        const float* arow = s_row + 32 * m * S;
#pragma unroll
        for (int qb = 0; qb < KS / 4; ++qb) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * qb);
#if LFGC_ABLATE & 4
            acc[qb & 15] += a4.x * Bin[4 * qb] + a4.y * Bin[4 * qb + 1] + a4.z * Bin[4 * qb + 2] + a4.w * Bin[4 * qb + 3];
#else
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, Bin[4 * qb + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, Bin[4 * qb + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, Bin[4 * qb + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, Bin[4 * qb + 3], acc, 0, 0, 0);
#endif
        }
        if (STASH) {
            // one opaque row pointer per tile + immediate offsets (r * 256 B): otherwise hipcc hoists a 64-bit
            // address per store out of the batch loop and spills ~60 VGPRs
            float* pm = stash + m * (16 * 64) + lane;
            asm volatile("" : "+v"(pm));
#pragma unroll
            for (int r = 0; r < 16; ++r) pm[r * 64] = acc[r];
        }
        float amax = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) amax = lfgc_absmax3(amax, acc[r], acc[r + 1]);
#if LFGC_ABLATE & 2
#pragma unroll
        for (int r = 0; r < 16; ++r) Bout[16 * m + r] = 0.5f * acc[r];
        amax = 0.0f;
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) Bout[16 * m + r] = lfgc_snake_t<false>(acc[r]);
#endif
        if (__builtin_expect(__any(amax > LFGC_TRIG_FAST_MAX), 0)) {       // wave-uniform; a diverged model only
#pragma unroll
            for (int r = 0; r < 16; ++r) Bout[16 * m + r] = lfgc_snake_t<true>(acc[r]);
        }
    }
}

// STREAM = false: WAVES = 4, two workgroups per CU, layer blocks staged once ("resident" nets that fit 80 KB of LDS).
// STREAM = true : one workgroup per CU, the layer blocks stream through a 2-deep LDS ring by LDS-DMA: the block of
//            layer t+1 is in flight while layer t computes, one barrier per layer, no exposed staging.  WAVES = 8
//            (256 samples per batch) when there are enough batches to give every CU one, else WAVES = 4 so that a
//            32 768-sample call (one reference tile / train step) still spreads over all 256 CUs.
template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH>
__global__ __launch_bounds__(WAVES * 64, 2) void lfgc_fwd_kernel(const LfgcFwdArgs a) {
    constexpr int E = 3 + 6 * NF;
    constexpr int EP = (E + 7) / 8 * 8;
    constexpr int K0P = CH + EP;
    constexpr int KS0 = K0P / 2;
    constexpr int HP = 32 * MT;
    constexpr int KS1 = HP / 2;
    constexpr int S0 = K0P + 4;
    constexpr int S1 = HP + 4;
    constexpr int BLK0 = HP * S0 + HP;
    constexpr int BLK1 = HP * S1 + HP;
    constexpr int BLKMAX = BLK0 > BLK1 ? BLK0 : BLK1;
    constexpr int CHH = CH / 2;          // channels gathered per lane
    constexpr int EPH = EP / 2;          // scalar inputs carried per lane
    constexpr int NT = WAVES * 64;

    // range fallback of the f16 builds: nothing to redo (uniform).  Device-scope load: the word is written by the kernel
    // enqueued just before this one (and cleared by a memset before that), possibly through another XCD's L2
    if (a.redo_if && __hip_atomic_load(a.redo_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_final = smem;               // Wf (HP) | bf (4)
    float* s_w = smem + HP + 4;          // resident: every layer block; streamed: ring of 2 x BLKMAX
    // lattice mode: coordinate of every voxel index per axis, built once per workgroup (the per-sample form
    // costs two fp64 divisions per axis); placed behind the weight region
    float* s_coord = s_w + (STREAM ? 2 * BLKMAX : (BLK0 + (a.L - 1) * BLK1));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int hh = lane >> 5;
    const int L = a.L;
    const int off_final = BLK0 + (L - 1) * BLK1;

    {   // final layer (+ every layer block when resident): staged once per workgroup
        const f32x4* src = reinterpret_cast<const f32x4*>(a.packed + off_final);
        for (int i = tid; i < (HP + 4) / 4; i += NT) reinterpret_cast<f32x4*>(s_final)[i] = src[i];
        if (!STREAM) {
            const f32x4* srcw = reinterpret_cast<const f32x4*>(a.packed);
            for (int i = tid; i < off_final / 4; i += NT) reinterpret_cast<f32x4*>(s_w)[i] = srcw[i];
        } else {
            lfgc_dma_to_lds(a.packed, s_w, BLK0, wave, lane, WAVES);      // layer 0 of the first batch -> slot 0
        }
    }
    if (!a.pos && a.coord_table) {
        const int r01 = a.res0 + a.res1, r012 = r01 + a.res2;
        for (int i = tid; i < r012; i += NT) {
            s_coord[i] = i < a.res0 ? lfgc_lattice_coord(i, a.res0, a.tile, a.scale0)
                       : i < r01 ? lfgc_lattice_coord(i - a.res0, a.res1, a.tile, a.scale1)
                                 : lfgc_lattice_coord(i - r01, a.res2, a.tile, a.scale2);
        }
    }
    __syncthreads();
    unsigned step = 0;                   // streamed: layers executed so far (ring slot = step & 1)

    const long long N = a.n;
    for (long long batch = blockIdx.x; batch < a.nbatches; batch += gridDim.x) {
        const long long tile_idx = batch * WAVES + wave;
        const long long n = tile_idx * LFGC_TILE_SAMPLES + j;
        const bool valid = n < N;
        const long long nc = valid ? n : (N - 1);

        float B0[KS0];
        lfgc_sample_inputs<CH, NF>(a, nc, N, s_coord, hh, B0);

        float* stash_tile = nullptr;
        if (STASH) {
            stash_tile = a.stash + tile_idx * (long long)(64 * (KS0 + L * 16 * MT));
            {
                float* px = stash_tile + lane;
                asm volatile("" : "+v"(px));
#pragma unroll
                for (int s = 0; s < KS0; ++s) px[s * 64] = B0[s];
            }
            stash_tile += 64 * KS0;
        }

        // ---- layers ------------------------------------------------------------------------------------
        // streamed: this layer's block was put in flight one step ago: wait for my pieces, then for everyone's;
        // the same barrier says every wave is done with the other ring slot -> prefetch the next block into it
        auto acquire = [&](int l) -> const float* {
            if (!STREAM) return s_w + (l == 0 ? 0 : BLK0 + (l - 1) * BLK1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const float* blk = s_w + (step & 1) * BLKMAX;
            const int ln = (l + 1 == L) ? 0 : l + 1;
            if (ln != 0 || batch + gridDim.x < a.nbatches) {
                const float* src = a.packed + (ln == 0 ? 0 : BLK0 + (long long)(ln - 1) * BLK1);
                lfgc_dma_to_lds(src, s_w + ((step + 1) & 1) * BLKMAX, ln == 0 ? BLK0 : BLK1, wave, lane, WAVES);
            }
            ++step;
            return blk;
        };
        float Bn[16 * MT];
        {
            const float* blk = acquire(0);
            lfgc_layer_fwd<KS0, MT, S0, STASH>(blk, B0, Bn, stash_tile, j, hh, lane);
        }
        // hidden layers two at a time, ping-ponging between two register arrays (no per-layer copy)
        {
            float Bm[16 * MT];
            int l = 1;
            for (; l + 1 < L; l += 2) {
                const float* blk = acquire(l);
                lfgc_layer_fwd<KS1, MT, S1, STASH>(blk, Bn, Bm, STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr,
                                                   j, hh, lane);
                blk = acquire(l + 1);
                lfgc_layer_fwd<KS1, MT, S1, STASH>(blk, Bm, Bn, STASH ? stash_tile + (long long)(l + 1) * (64 * 16 * MT) : nullptr,
                                                   j, hh, lane);
            }
            if (l < L) {
                const float* blk = acquire(l);
                lfgc_layer_fwd<KS1, MT, S1, STASH>(blk, Bn, Bm, STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr,
                                                   j, hh, lane);
#pragma unroll
                for (int s = 0; s < KS1; ++s) Bn[s] = Bm[s];
            }
        }

        // ---- final Linear (H -> 1): per-lane partial dot + exchange between the two lane halves ------
        float y = 0.0f;
#pragma unroll
        for (int qb = 0; qb < KS1 / 4; ++qb) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_final + 8 * qb + 4 * hh);
            y = __builtin_fmaf(w4.x, Bn[4 * qb + 0], y);
            y = __builtin_fmaf(w4.y, Bn[4 * qb + 1], y);
            y = __builtin_fmaf(w4.z, Bn[4 * qb + 2], y);
            y = __builtin_fmaf(w4.w, Bn[4 * qb + 3], y);
        }
        y += __shfl_xor(y, 32);
        y += s_final[HP];
        if (a.clamp) y = fminf(fmaxf(y, -1.0f), 1.0f);
        if (valid && hh == 0) a.out[n] = y;
    }
}

// Host-side launcher for one (CH, MT) pair: picks the resident (4-wave) or streamed (8-wave) build and the
// stash / no-stash build.
template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH>
static int lfgc_launch_fwd_one(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    auto kern = lfgc_fwd_kernel<CH, MT, NF, WAVES, STREAM, STASH>;
    static int lds_limit_set[LFGC_MAX_DEVICES] = {0};   // per (instantiation, device); raised once (launches stay graph-capturable)
    const int dev = lfgc_current_device();
    if (lds_bytes > 64 * 1024 && lds_bytes > lds_limit_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        lds_limit_set[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

template <int CH, int MT, int NF>
static int lfgc_launch_fwd(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    if (a.resident) {
        return a.stash ? lfgc_launch_fwd_one<CH, MT, NF, 4, false, true>(a, lds_bytes, grid, stream)
                       : lfgc_launch_fwd_one<CH, MT, NF, 4, false, false>(a, lds_bytes, grid, stream);
    }
    if (a.waves == 8) {
        return a.stash ? lfgc_launch_fwd_one<CH, MT, NF, 8, true, true>(a, lds_bytes, grid, stream)
                       : lfgc_launch_fwd_one<CH, MT, NF, 8, true, false>(a, lds_bytes, grid, stream);
    }
    return a.stash ? lfgc_launch_fwd_one<CH, MT, NF, 4, true, true>(a, lds_bytes, grid, stream)
                   : lfgc_launch_fwd_one<CH, MT, NF, 4, true, false>(a, lds_bytes, grid, stream);
}
