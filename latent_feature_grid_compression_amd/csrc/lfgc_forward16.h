// lfgc_forward16.h -- the fused forward with the MLP contractions on the f16 matrix pipe at fp32-level accuracy.
//
// Same boundary, mapping and data flow as lfgc_forward.h (replaces model/Feature_Grid_Model.py:62-78), but every
// fp32 operand of the layer GEMMs is carried as an f16 pair x = hi + lo (hi = rne_f16(x), lo = rne_f16(x - hi):
// 22-24 significant bits) and each fp32 product block is three v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//        W.h  ~=  W_hi.h_hi + W_hi.h_lo + W_lo.h_hi            (the dropped W_lo.h_lo term is <= 2^-22 relative)
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the f16 rate (measured 68 cycles per 4 kFLOP vs 34 cycles per 32
// kFLOP, tools/microbench/mfma_interleave.hip), so three f16 MFMAs do the work of eight f32 ones in 96 instead of
// 512 pipe cycles.  Measured end-to-end error vs the reference is at the level of the exact-fp32 build (fp32
// accumulation order dominates; tests/test_hip_forward.py).
// Range: weights are pre-scaled per layer by a power of two (pack_scale_kernel) so that hi/lo halves sit in the
// middle of the f16 range; the accumulator is scaled back exactly.  Activations must stay below 65504 in magnitude
// (f16 overflow -> inf/NaN); the exact build (precision 0) has no such limit.
//
// Register mapping: one wave = 32 samples; D/C layout of the 32x32x16 MFMA equals the 32x32x2 one, so the chain
// "accumulators -> SnakeAlt -> next layer's B operand" is kept: accumulator registers 8s..8s+7 of tile m become, as
// an f16x8 fragment, exactly the B operand of k-step 2m + s (the weight columns are stored in that k order).
#pragma once
#include "lfgc_forward.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x[0..8) fp32 -> hi / lo f16 fragments, 2 VALU instructions per value: one packed RNE conversion per pair for each
// of hi and lo, and the remainder x - f32(hi) as ONE mixed-precision FMA per value (v_fma_mix_f32 reads the f16 half
// in place: hi * -1 + x, exact).  (Plain C++ casts cost 4 per value: hipcc converts every hi half twice.)
__device__ __forceinline__ void lfgc_split8(const float* __restrict__ x, h16x8& hi, h16x8& lo) {
    u32x4 hp, lp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        unsigned h, l;
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x[2 * t]), "v"(x[2 * t + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x[2 * t]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x[2 * t + 1]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
        hp[t] = h;
        lp[t] = l;
    }
    hi = __builtin_bit_cast(h16x8, hp);
    lo = __builtin_bit_cast(h16x8, lp);
}

// Reduced-precision build (LFGC_PRECISION_F16): the hi half only, one packed conversion per pair.
__device__ __forceinline__ void lfgc_cvt8(const float* __restrict__ x, h16x8& hi) {
    u32x4 hp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        unsigned h;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x[2 * t]), "v"(x[2 * t + 1]));
        hp[t] = h;
    }
    hi = __builtin_bit_cast(h16x8, hp);
}

// One hidden layer on a 32-sample tile.  LAST = false: outputs the next layer's fragments; LAST = true: folds the
// final Linear (H -> 1) in and returns this lane's partial dot product through `ydot`.
// SPLIT = false: single f16 product W_hi.h_hi (the lo halves of the weight images and of the activations are ignored).
template <int KS16, int MT, int S, bool STASH, bool LAST, bool SPLIT>
__device__ __forceinline__ void lfgc_layer_fwd16(const float* __restrict__ s_blk, const h16x8 (&Bhi)[KS16],
                                                 const h16x8 (&Blo)[KS16], float inv_scale,
                                                 h16x8 (&Ohi)[2 * MT], h16x8 (&Olo)[2 * MT],
                                                 const float* __restrict__ s_final, float& ydot,
                                                 float* __restrict__ stash, int j, int hh, int lane) {
    const float* s_bias = s_blk + 32 * MT * S + 4 * hh;
    const float* s_row = s_blk + j * S + 8 * hh;              // lane half hh: bytes [32 hh, 32 hh + 32) of each 64-B k-step
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
        for (int ks = 0; ks < KS16; ++ks) {
            const h16x8 whi = *reinterpret_cast<const h16x8*>(arow + 16 * ks);
#if LFGC_ABLATE & 4
            acc[ks & 15] += (float)whi[0] * (float)Bhi[ks][0];
#else
            if (SPLIT) {
                const h16x8 wlo = *reinterpret_cast<const h16x8*>(arow + 16 * ks + 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, Bhi[ks], acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, Blo[ks], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, Bhi[ks], acc, 0, 0, 0);
#endif
        }
        float av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) av[r] = acc[r] * inv_scale;          // exact: power of two
        if (STASH) {
            float* pm = stash + m * (16 * 64) + lane;
            asm volatile("" : "+v"(pm));
#pragma unroll
            for (int r = 0; r < 16; ++r) pm[r * 64] = av[r];
        }
        // no range screen here (the exact build has one): |a| > 2^15 means |h| ~ |a|/2 is about to leave the f16
        // range this build requires anyway; the polynomial path stays finite and degrades gracefully up to there
        float hv[16];
#if LFGC_ABLATE & 2
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = 0.5f * av[r];
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = lfgc_snake_t<false>(av[r]);
#endif
        if (LAST) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_final + 32 * m + 8 * q + 4 * hh);
                ydot = __builtin_fmaf(w4.x, hv[4 * q + 0], ydot); ydot = __builtin_fmaf(w4.y, hv[4 * q + 1], ydot);
                ydot = __builtin_fmaf(w4.z, hv[4 * q + 2], ydot); ydot = __builtin_fmaf(w4.w, hv[4 * q + 3], ydot);
            }
        } else {
#if LFGC_ABLATE & 8
#pragma unroll
            for (int t = 0; t < 8; ++t) { Ohi[2 * m][t] = (_Float16)hv[t]; Olo[2 * m][t] = (_Float16)0; Ohi[2 * m + 1][t] = (_Float16)hv[8 + t]; Olo[2 * m + 1][t] = (_Float16)0; }
#else
            if (SPLIT) {
                lfgc_split8(hv, Ohi[2 * m], Olo[2 * m]);
                lfgc_split8(hv + 8, Ohi[2 * m + 1], Olo[2 * m + 1]);
            } else {
                lfgc_cvt8(hv, Ohi[2 * m]);
                lfgc_cvt8(hv + 8, Ohi[2 * m + 1]);
            }
#endif
        }
    }
}

template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH, bool SPLIT>
__global__ __launch_bounds__(WAVES * 64, 2) void lfgc_fwd16_kernel(const LfgcFwdArgs a) {
    constexpr int E = 3 + 6 * NF;
    constexpr int EP = (E + 7) / 8 * 8;
    constexpr int K0P = CH + EP;
    constexpr int KS0 = K0P / 2;                 // fp32 inputs per lane (stash layout of the exact build)
    constexpr int K0P16 = (K0P + 15) / 16 * 16;
    constexpr int KS16_0 = K0P16 / 16;           // 16-wide k-steps of layer 0
    constexpr int HP = 32 * MT;
    constexpr int KS16_1 = HP / 16;
    constexpr int S0 = K0P16 + 4;
    constexpr int S1 = HP + 4;
    constexpr int BLK0 = HP * S0 + HP;
    constexpr int BLK1 = HP * S1 + HP;
    constexpr int BLKMAX = BLK0 > BLK1 ? BLK0 : BLK1;
    constexpr int NT = WAVES * 64;
    // offsets inside the packed blob (lfgc_common.h)
    constexpr int F_BLK0 = HP * (K0P + 4) + HP, F_BLK1 = HP * (HP + 4) + HP;
    constexpr int K0R = (K0P + 31) / 32 * 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_final = smem;               // Wf (HP) | bf (4)
    float* s_scale = smem + HP + 4;      // scale[8] | 1/scale[8]
    float* s_w = s_scale + 16;           // resident: every layer block; streamed: ring of 2 x BLKMAX
    float* s_coord = s_w + (STREAM ? 2 * BLKMAX : (BLK0 + (a.L - 1) * BLK1));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int hh = lane >> 5;
    const int L = a.L;
    const int off_final = F_BLK0 + (L - 1) * F_BLK1;
    const int off_h = off_final + HP + 4 + K0R * (HP + 4) + (L - 1) * HP * (HP + 4);
    const float* hblk = a.packed + off_h + 16;

    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.packed + off_final);
        for (int i = tid; i < (HP + 4) / 4; i += NT) reinterpret_cast<f32x4*>(s_final)[i] = src[i];
        if (tid < 16) s_scale[tid] = a.packed[off_h + tid];
        if (!STREAM) {
            const f32x4* srcw = reinterpret_cast<const f32x4*>(hblk);
            for (int i = tid; i < (BLK0 + (L - 1) * BLK1) / 4; i += NT) reinterpret_cast<f32x4*>(s_w)[i] = srcw[i];
        } else {
            lfgc_dma_to_lds(hblk, s_w, BLK0, wave, lane, WAVES);
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
    unsigned step = 0;

    const long long N = a.n;
    for (long long batch = blockIdx.x; batch < a.nbatches; batch += gridDim.x) {
        const long long tile_idx = batch * WAVES + wave;
        const long long n = tile_idx * LFGC_TILE_SAMPLES + j;
        const bool valid = n < N;
        const long long nc = valid ? n : (N - 1);

        float X[8 * KS16_0];
        {
            float B0[KS0];
            lfgc_sample_inputs<CH, NF>(a, nc, N, s_coord, hh, B0);
#pragma unroll
            for (int s = 0; s < KS0; ++s) X[s] = B0[s];
#pragma unroll
            for (int s = KS0; s < 8 * KS16_0; ++s) X[s] = 0.0f;
        }

        float* stash_tile = nullptr;
        if (STASH) {
            stash_tile = a.stash + tile_idx * (long long)(64 * (KS0 + L * 16 * MT));
            {
                float* px = stash_tile + lane;
                asm volatile("" : "+v"(px));
#pragma unroll
                for (int s = 0; s < KS0; ++s) px[s * 64] = X[s];
            }
            stash_tile += 64 * KS0;
        }

        auto acquire = [&](int l) -> const float* {
            if (!STREAM) return s_w + (l == 0 ? 0 : BLK0 + (l - 1) * BLK1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const float* blk = s_w + (step & 1) * BLKMAX;
            const int ln = (l + 1 == L) ? 0 : l + 1;
            if (ln != 0 || batch + gridDim.x < a.nbatches) {
                const float* src = hblk + (ln == 0 ? 0 : BLK0 + (long long)(ln - 1) * BLK1);
                lfgc_dma_to_lds(src, s_w + ((step + 1) & 1) * BLKMAX, ln == 0 ? BLK0 : BLK1, wave, lane, WAVES);
            }
            ++step;
            return blk;
        };

        float ydot = 0.0f;
        h16x8 Ahi[2 * MT], Alo[2 * MT], Bhi[2 * MT], Blo[2 * MT];
        {   // layer 0
            h16x8 X0hi[KS16_0], X0lo[KS16_0];
#pragma unroll
            for (int s = 0; s < KS16_0; ++s) {
                if (SPLIT) lfgc_split8(X + 8 * s, X0hi[s], X0lo[s]);
                else lfgc_cvt8(X + 8 * s, X0hi[s]);
            }
            const float* blk = acquire(0);
            if (L == 1)
                lfgc_layer_fwd16<KS16_0, MT, S0, STASH, true, SPLIT>(blk, X0hi, X0lo, s_scale[8], Ahi, Alo, s_final, ydot, stash_tile, j, hh, lane);
            else
                lfgc_layer_fwd16<KS16_0, MT, S0, STASH, false, SPLIT>(blk, X0hi, X0lo, s_scale[8], Ahi, Alo, s_final, ydot, stash_tile, j, hh, lane);
        }
        // hidden layers 1 .. L-2 in ping-pong pairs, then the last one with the head folded in
        {
            int l = 1;
            for (; l + 2 < L; l += 2) {
                const float* blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, false, SPLIT>(blk, Ahi, Alo, s_scale[8 + l], Bhi, Blo, s_final, ydot,
                                                              STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr, j, hh, lane);
                blk = acquire(l + 1);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, false, SPLIT>(blk, Bhi, Blo, s_scale[9 + l], Ahi, Alo, s_final, ydot,
                                                              STASH ? stash_tile + (long long)(l + 1) * (64 * 16 * MT) : nullptr, j, hh, lane);
            }
            if (l + 1 < L) {       // one more non-final layer: A -> B, final consumes B
                const float* blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, false, SPLIT>(blk, Ahi, Alo, s_scale[8 + l], Bhi, Blo, s_final, ydot,
                                                              STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr, j, hh, lane);
                ++l;
                blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, true, SPLIT>(blk, Bhi, Blo, s_scale[8 + l], Ahi, Alo, s_final, ydot,
                                                             STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr, j, hh, lane);
            } else if (l < L) {    // final layer consumes A
                const float* blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, true, SPLIT>(blk, Ahi, Alo, s_scale[8 + l], Bhi, Blo, s_final, ydot,
                                                             STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr, j, hh, lane);
            }
        }

        float y = ydot + __shfl_xor(ydot, 32);
        y += s_final[HP];
        if (a.clamp) y = fminf(fmaxf(y, -1.0f), 1.0f);
        if (valid && hh == 0) a.out[n] = y;
    }
}

template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH, bool SPLIT>
static int lfgc_launch_fwd16_cfg(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    auto kern = lfgc_fwd16_kernel<CH, MT, NF, WAVES, STREAM, STASH, SPLIT>;
    static int lds_limit_set = 0;          // per instantiation; raised once (also keeps launches graph-capturable)
    if (lds_bytes > 64 * 1024 && lds_bytes > lds_limit_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        lds_limit_set = lds_bytes;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH>
static int lfgc_launch_fwd16_one(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    return a.single ? lfgc_launch_fwd16_cfg<CH, MT, NF, WAVES, STREAM, STASH, false>(a, lds_bytes, grid, stream)
                    : lfgc_launch_fwd16_cfg<CH, MT, NF, WAVES, STREAM, STASH, true>(a, lds_bytes, grid, stream);
}

template <int CH, int MT, int NF>
static int lfgc_launch_fwd16(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    if (a.resident) {
        return a.stash ? lfgc_launch_fwd16_one<CH, MT, NF, 4, false, true>(a, lds_bytes, grid, stream)
                       : lfgc_launch_fwd16_one<CH, MT, NF, 4, false, false>(a, lds_bytes, grid, stream);
    }
    if (a.waves == 8) {
        return a.stash ? lfgc_launch_fwd16_one<CH, MT, NF, 8, true, true>(a, lds_bytes, grid, stream)
                       : lfgc_launch_fwd16_one<CH, MT, NF, 8, true, false>(a, lds_bytes, grid, stream);
    }
    return a.stash ? lfgc_launch_fwd16_one<CH, MT, NF, 4, true, true>(a, lds_bytes, grid, stream)
                   : lfgc_launch_fwd16_one<CH, MT, NF, 4, true, false>(a, lds_bytes, grid, stream);
}
