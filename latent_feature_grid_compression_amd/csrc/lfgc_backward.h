// lfgc_backward.h -- backward of the fused sample + embed + MLP path for gfx950 (exact fp32 MFMA).
//
// What autograd derives for model/Feature_Grid_Model.py:62-75 in the reference (triggered at
// training/training.py:137), as three kernels:
//   1. lfgc_bwd_data_kernel   per 32-sample wave tile, the forward's register mapping run backwards:
//        dH_L = Wf dy ; for l = L..1: dA_l = dH_l * snake'(a_l) (a_l from the forward's stash),
//        dH_{l-1} = W_l^T dA_l  (A operand = transposed weight image in LDS, B operand = dA_l in VGPRs,
//        the accumulators are again the next step's B operand), dX0 = W_0^T dA_1 lands in exactly the
//        registers the forward's layer-0 input occupied: grid-feature gradients -> float-atomic scatter into
//        the channel-last d_grid (staged through LDS so that one atomic wave-instruction covers whole 128-B
//        channel rows), scalar-input gradients (+ the sampler's coordinate gradient) -> d_pos.
//        dA_l is also written to a scratch "dstash" for kernel 2.
//   2. lfgc_bwd_weight_kernel  dW_l = dA_l^T H_{l-1}: contraction over SAMPLES (K = 32 per tile), M = h_out,
//        N = k_in; operands are read back from stash/dstash so that 16 consecutive samples of one row are one
//        64-byte load per lane; each workgroup accumulates its share of the sample tiles in registers and
//        writes one partial slab.
//   3. lfgc_bwd_reduce_kernel  sums the slabs into the nn.Linear-shaped gradients (deterministic, no atomics).
#pragma once
#include "lfgc_common.h"
#include "lfgc_forward16.h"      // h16x8, lfgc_split8

struct LfgcBwdArgs {
    const float* pos;          // (N,3)
    long long n;
    const float* grid;         // (D,H,W,Cs)   (read only when d_pos is requested)
    int D, H, W, Cs;
    const float* packed;
    int L;
    const float* stash;
    const float* d_out;        // (N)
    float* dstash;             // [tiles][L*16*MT][64]
    float* dscale;             // [tiles][L]: the power-of-two scale each tile's dA_l was split with (f16 builds), for the weight kernel
    float* d_grid;             // (D,H,W,Cs), accumulated with float atomics
    float* dfeat;              // nullptr: the data kernel scatters into d_grid itself; else (tiles*32, CH) scratch: it only
                               // writes the feature gradients there and lfgc_bwd_scatter_kernel does the atomics
    float* d_pos;              // (N,3) or nullptr
    long long nbatches;
    unsigned long long* stamps;   // diagnostics builds (-DLFGC_STAMPS, tools/phase_stamps.py bwd): per-wave cycle totals per phase
};

#ifdef LFGC_STAMPS
#define LFGC_BSTAMP(k) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); bst[k] += now__ - bst_last; bst_last = now__; } while (0)
#else
#define LFGC_BSTAMP(k) do { } while (0)
#endif

// acc += W_tile . B for one 32-row M tile; `arow` = LDS address of (row 32m + lane&31, column 4*(lane>>5)).
template <int KS>
__device__ __forceinline__ f32x16 lfgc_mfma_tile(const float* __restrict__ arow, const float (&Bin)[KS], f32x16 acc) {
    static_assert(KS % 4 == 0, "k-steps come in groups of 4 (one ds_read_b128)");
#pragma unroll
    for (int qb = 0; qb < KS / 4; ++qb) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * qb);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, Bin[4 * qb + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, Bin[4 * qb + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, Bin[4 * qb + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, Bin[4 * qb + 3], acc, 0, 0, 0);
    }
    return acc;
}

// dA[i] = dH[i] * snake'(a[i]) for the 16*MT pre-activations of one layer (a read from the stash slot).
template <int MT>
__device__ __forceinline__ void lfgc_snake_bwd(const float* __restrict__ slot, const float (&dH)[16 * MT],
                                               float (&dA)[16 * MT], int lane) {
    float av[16 * MT];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16 * MT; ++i) {
        av[i] = slot[i * 64 + lane];
        bad |= lfgc_trig_out_of_range(av[i]);
    }
#pragma unroll
    for (int i = 0; i < 16 * MT; ++i) dA[i] = dH[i] * lfgc_snake_grad_t<false>(av[i]);
    if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
        for (int i = 0; i < 16 * MT; ++i) dA[i] = dH[i] * lfgc_snake_grad_t<true>(av[i]);
    }
}

// One workgroup per CU (WAVES = 8 when every CU gets a 256-sample batch, else 4); the transposed weight images
// stream through a 2-deep LDS ring by LDS-DMA exactly like the forward's (image of step t+1 in flight while step
// t computes, one barrier per step); the scatter staging aliases the ring slot that has just been consumed.
// acc += Wt_tile . dA for one 32-row tile with f16-split operands (three MFMAs per 16-wide k-step); `arow` = LDS
// address of (row 32m + lane&31, lane half's 32 bytes of k-step 0).
// `dma`: the next image's weight stream (lfgc_dma_piece, lfgc_common.h); pieces [pi0, pi0 + KS16) capped at NPW go out one
// per k-step, in the shadow of its MFMAs, instead of all at once after the layer's barrier (NPW = 0: not streamed here).
template <int KS16, bool SPLIT, int NPW = 0, int WAVES = 4>
__device__ __forceinline__ f32x16 lfgc_mfma_tile16(const float* __restrict__ arow, const h16x8 (&Fhi)[KS16],
                                                   const h16x8 (&Flo)[KS16], f32x16 acc,
                                                   const LfgcDmaPlan* dma = nullptr, int pi0 = 0) {
#pragma unroll
    for (int ks = 0; ks < KS16; ++ks) {
        if constexpr (NPW > 0) { if (pi0 + ks < NPW) lfgc_dma_piece<WAVES>(*dma, pi0 + ks); }
        const h16x8 whi = *reinterpret_cast<const h16x8*>(arow + 16 * ks);
        if (SPLIT) {
            const h16x8 wlo = *reinterpret_cast<const h16x8*>(arow + 16 * ks + 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, Fhi[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, Flo[ks], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, Fhi[ks], acc, 0, 0, 0);   // SPLIT = false: the single product
    }
    return acc;
}

// Gradients span many orders of magnitude and are small: before the f16 hi/lo split each wave scales its 32-sample
// tile of dA by a power of two that brings the tile maximum to [2^13, 2^14) (so that the lo halves, 2^-11 of the hi,
// do not fall off the bottom of the f16 range); the product is scaled back exactly.  Returns the scale, writes 1/scale.
template <int NV>
__device__ __forceinline__ float lfgc_tile_pow2_scale(const float (&v)[NV], float& inv) {
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; i += 2) amax = lfgc_absmax3(amax, v[i], v[i + 1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    const int e = (__float_as_int(amax) >> 23) & 255;            // amax = 1.x * 2^(e - 127); e == 0: zero tile
    int k = (e == 0 || e == 255) ? 0 : (13 + 127 - e);
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    inv = __int_as_float((127 - k) << 23);
    return __int_as_float((127 + k) << 23);
}

template <int NF16, bool SPLIT>
__device__ __forceinline__ void lfgc_split_scaled(const float* __restrict__ v, float sc, h16x8 (&Fhi)[NF16], h16x8 (&Flo)[NF16]) {
#pragma unroll
    for (int f = 0; f < NF16; ++f) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = v[8 * f + u] * sc;
        if (SPLIT) lfgc_split8(t, Fhi[f], Flo[f]);
        else lfgc_cvt8(t, Fhi[f]);
    }
}

// PREC (the C-ABI precision code) 0: exact f32 MFMA chain.  1: the chain's GEMMs run f16-split like the default forward
// build (dA carried as f16 hi+lo fragments, transposed weight images pre-split and scaled; fp32 accumulate, scaled
// back).  2: reduced precision, the hi halves only (one f16 product).
template <int CH, int MT, int NF, int WAVES, int PREC>
__global__ __launch_bounds__(WAVES * 64, 2) void lfgc_bwd_data_kernel(const LfgcBwdArgs a) {
    constexpr bool H16 = PREC != 0;
    constexpr bool SPLIT = PREC == 1;
    constexpr int E = 3 + 6 * NF;
    constexpr int EP = (E + 7) / 8 * 8;
    constexpr int K0P = CH + EP;
    constexpr int K0R = (K0P + 31) / 32 * 32;
    constexpr int KS0 = K0P / 2;
    constexpr int HP = 32 * MT;
    constexpr int KS1 = HP / 2;
    constexpr int S0 = K0P + 4, S1 = HP + 4, ST = HP + 4;
    constexpr int BLK0 = HP * S0 + HP, BLK1 = HP * S1 + HP;
    constexpr int TB0 = K0R * ST, TB1 = HP * ST;
    constexpr int CHH = CH / 2, EPH = EP / 2;
    constexpr int TXF = (CHH + 15) / 16;         // M tiles of dX0 that hold grid-feature gradients
    constexpr int TXA = K0R / 32;                // ... all of dX0 (features + scalar inputs)
    constexpr int SCS = CH + 4;                  // scatter staging row stride (floats)
    constexpr int SC_WAVE = 32 * (SCS + 16);     // per wave: dfeat rows + 8 weights + 8 offsets per sample
    constexpr int SPI = 64 / CH;                 // samples covered by one atomic wave-instruction
    constexpr int TBMAX = TB0 > TB1 ? TB0 : TB1;
    constexpr int SLOT = TBMAX > WAVES * SC_WAVE ? TBMAX : WAVES * SC_WAVE;   // ring slot (floats)
    constexpr int NT = WAVES * 64;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_final = smem;               // Wf (HP) | bf (4)
    float* s_inv = smem + HP + 4;        // 1 / scale per layer (f16-split build), 8 floats
    float* s_ring = s_inv + 8;           // 2 x SLOT: transposed weight images / scatter staging

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int L = a.L;
    const int off_final = BLK0 + (L - 1) * BLK1;
    const int off_t = off_final + HP + 4;
    const long long per_tile = 64LL * (KS0 + L * 16 * MT);
    const long long dper_tile = 64LL * (L * 16 * MT);

    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.packed + off_final);
        for (int i = tid; i < (HP + 4) / 4; i += NT) reinterpret_cast<f32x4*>(s_final)[i] = src[i];
    }
    // image sequence per batch: W_{L-1}^T, ..., W_1^T (TB1 each), then W_0^T (TB0)
    constexpr int K0P16 = (K0P + 15) / 16 * 16;
    constexpr int HBLK0 = HP * (K0P16 + 4) + HP, HBLK1 = HP * (HP + 4) + HP;
    const int off_h = off_t + TB0 + (L - 1) * TB1;            // scales, then the f16-split forward blocks (lfgc_common.h)
    const int off_img = H16 ? off_h + 32 + LFGC_MAX_LAYERS * HP + HP + HBLK0 + (L - 1) * HBLK1 : off_t;
    auto image_src = [&](int l) -> const float* {
        return l == 0 ? a.packed + off_img : a.packed + off_img + TB0 + (long long)(l - 1) * TB1;
    };
    if (tid < LFGC_MAX_LAYERS) s_inv[tid] = a.packed[off_h + 8 + tid];      // all LDS lives in the one dynamic array
    lfgc_dma_to_lds(image_src(L - 1), s_ring, (L - 1) == 0 ? TB0 : TB1, wave, lane, WAVES);
    __syncthreads();
    unsigned step = 0;
#ifdef LFGC_STAMPS
    unsigned long long bst[8] = {0};
    unsigned long long bst_last = __builtin_amdgcn_s_memtime();
    const unsigned long long bst_t0 = bst_last;
#endif

#ifndef LFGC_BWD_DMA_BURST
#define LFGC_BWD_DMA_BURST 0         // diagnostics: 1 = every wave issues its pieces back to back after the barrier (round 2)
#endif
    constexpr int NPW = (H16 && !LFGC_BWD_DMA_BURST) ? ((TBMAX / 4 + 63) / 64 + WAVES - 1) / WAVES : 0;   // pieces per wave and image
    LfgcDmaPlan dma = {a.packed, s_ring, TB1 / 4, __builtin_amdgcn_readfirstlane(wave), (unsigned)lane * 16u, 0ull, 0u};
    const long long N = a.n;
    for (long long batch = blockIdx.x; batch < a.nbatches; batch += gridDim.x) {
        asm volatile("" : "+s"(dma.wave));                // (keeps the pieces' address arithmetic inside the loop)
        // the image of layer l was put in flight one step ago; after the barrier every wave is also done with the
        // other slot, so the next image (layer l-1, or the next batch's first) goes into it
        auto acquire = [&](int l) -> const float* {
            LFGC_BSTAMP(1);                                   // snake' (stash loads) + dstash stores + scale + split
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            LFGC_BSTAMP(2);                                   // wait for stores / DMA + barrier
            const float* img = s_ring + (step & 1) * SLOT;
            const int ln = (l == 0) ? L - 1 : l - 1;
            if (H16 && !LFGC_BWD_DMA_BURST) {
                // streamed piece by piece under this step's MFMAs (always a real block: after the last batch the slot is
                // filled once more for nobody, which keeps the MFMA loops free of branches)
                dma.src = image_src(ln);
                dma.dst = s_ring + ((step + 1) & 1) * SLOT;
                dma.nvec = (ln == 0 ? TB0 : TB1) / 4;
            } else if (l != 0 || batch + gridDim.x < a.nbatches) {
                lfgc_dma_to_lds(image_src(ln), s_ring + ((step + 1) & 1) * SLOT, ln == 0 ? TB0 : TB1, wave, lane, WAVES);
            }
            ++step;
            return img;
        };
        const long long tile_idx = batch * WAVES + wave;
        const long long n = tile_idx * LFGC_TILE_SAMPLES + j;
        const bool valid = n < N;
        const long long nc = valid ? n : (N - 1);
        const float dy = valid ? a.d_out[n] : 0.0f;
        const float* st_tile = a.stash + tile_idx * per_tile;
        float* dst_tile = a.dstash + tile_idx * dper_tile;

        // ---- final Linear backward: dH_L[k] = Wf[k] * dy -------------------------------------------------
        float dH[16 * MT];
#pragma unroll
        for (int qb = 0; qb < KS1 / 4; ++qb) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_final + 8 * qb + 4 * hh);
            dH[4 * qb + 0] = w4.x * dy; dH[4 * qb + 1] = w4.y * dy;
            dH[4 * qb + 2] = w4.z * dy; dH[4 * qb + 3] = w4.w * dy;
        }

        LFGC_BSTAMP(0);
        // ---- hidden layers L-1 .. 1 (0-based): dA = dH * snake'(a), dH_prev = W^T dA ------------------------
        for (int l = L - 1; l >= 1; --l) {
            LFGC_BSTAMP(3);                                   // the MFMAs of the layer before (or the head's dH)
            float dA[16 * MT];
            lfgc_snake_bwd<MT>(st_tile + 64 * KS0 + (long long)l * (64 * 16 * MT), dH, dA, lane);
#pragma unroll
            for (int i = 0; i < 16 * MT; ++i) dst_tile[(long long)l * (64 * 16 * MT) + i * 64 + lane] = dA[i];
            if (H16) {
                h16x8 Fhi[2 * MT], Flo[2 * MT];
                float isc;
                const float sc = lfgc_tile_pow2_scale<16 * MT>(dA, isc);
                if (lane == 0) a.dscale[tile_idx * L + l] = sc;
                lfgc_split_scaled<2 * MT, SPLIT>(dA, sc, Fhi, Flo);
                const float* s_row = acquire(l) + j * ST + 8 * hh;
                const float is = s_inv[l] * isc;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    acc = lfgc_mfma_tile16<2 * MT, SPLIT, NPW, WAVES>(s_row + 32 * m * ST, Fhi, Flo, acc, &dma, m * 2 * MT);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dH[16 * m + r] = acc[r] * is;
                }
                if constexpr (NPW > MT * 2 * MT) {      // narrow nets: fewer k-steps than pieces (layer 0's image is the larger one)
                    for (int pi = MT * 2 * MT; pi < NPW; ++pi) lfgc_dma_piece<WAVES>(dma, pi);
                }
            } else {
                const float* s_row = acquire(l) + j * ST + 4 * hh;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    acc = lfgc_mfma_tile<KS1>(s_row + 32 * m * ST, dA, acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dH[16 * m + r] = acc[r];
                }
            }
        }

        // ---- layer 0: dA_1, then dX0 = W_0^T dA_1 in the forward's input register layout ------------------------
        float dX[16 * TXA];
        {
            float dA[16 * MT];
            lfgc_snake_bwd<MT>(st_tile + 64 * KS0, dH, dA, lane);
#pragma unroll
            for (int i = 0; i < 16 * MT; ++i) dst_tile[i * 64 + lane] = dA[i];
            h16x8 Fhi[2 * MT], Flo[2 * MT];
            float isc = 1.0f;
            if (H16) {
                const float sc = lfgc_tile_pow2_scale<16 * MT>(dA, isc);
                if (lane == 0) a.dscale[tile_idx * L] = sc;
                lfgc_split_scaled<2 * MT, SPLIT>(dA, sc, Fhi, Flo);
            }
            const float* s_row = acquire(0) + j * ST + (H16 ? 8 : 4) * hh;
            const float is = H16 ? s_inv[0] * isc : 1.0f;
#pragma unroll
            for (int m = 0; m < TXA; ++m) {
                if (m < TXF || a.d_pos) {            // scalar-input rows only when d_pos is wanted (wave-uniform)
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    if (H16) acc = lfgc_mfma_tile16<2 * MT, SPLIT, NPW, WAVES>(s_row + 32 * m * ST, Fhi, Flo, acc, &dma, m * 2 * MT);
                    else acc = lfgc_mfma_tile<KS1>(s_row + 32 * m * ST, dA, acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dX[16 * m + r] = acc[r] * is;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dX[16 * m + r] = 0.0f;
                }
            }
            if constexpr (NPW > 0) {          // pieces the (few, partly skipped) layer-0 tiles had no k-step for
                const int done = ((TXF < TXA && !a.d_pos) ? TXF : TXA) * 2 * MT;
                for (int pi = done; pi < NPW; ++pi) lfgc_dma_piece<WAVES>(dma, pi);
            }
        }

        LFGC_BSTAMP(3);
        // ---- sampler geometry (same arithmetic as the forward) ---------------------------------------------
        const float* pp = a.pos + 3 * nc;
        const float p0 = pp[0], p1 = pp[1], p2 = pp[2];
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(p0, 1.0f), (float)a.W), 1.0f), 2.0f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(p1, 1.0f), (float)a.H), 1.0f), 2.0f);
        const float iz = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(p2, 1.0f), (float)a.D), 1.0f), 2.0f);
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)a.W);
        const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)a.H);
        const int z0 = (int)fminf(fmaxf(fz0, -2.0f), (float)a.D);
        const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
        const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
        const float wz1 = __fsub_rn(iz, fz0), wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
        const bool in_range = (fx0 >= -1.0f) && (fx0 < (float)a.W) && (fy0 >= -1.0f) && (fy0 < (float)a.H) &&
                              (fz0 >= -1.0f) && (fz0 < (float)a.D);

        // ---- scatter d feat into d_grid ----------------------------------------------------------------------
        // in-kernel (throughput mode: other waves' work covers the atomics' latency): stage [sample][channel] +
        // per-corner weight / offset in LDS, then one atomic wave-instruction per 64 / CH samples and corner;
        // deferred (a.dfeat; small batches, one wave per SIMD: 128 dependent-latency atomics per wave were 40 % of
        // this kernel): only write the feature gradients out, lfgc_bwd_scatter_kernel scatters them at full occupancy
        const bool stage = a.dfeat == nullptr;            // uniform
        if (stage) __syncthreads();                       // every wave is done with the layer-0 image: reuse its slot
        float* s_df = s_ring + ((step - 1) & 1) * SLOT + wave * SC_WAVE;   // [32][SCS]
        float* s_cw = s_df + 32 * SCS;                    // [32][8] corner weights
        int* s_co = reinterpret_cast<int*>(s_cw + 32 * 8);   // [32][8] corner row offsets (floats)
#pragma unroll
        for (int c4 = 0; c4 < CHH / 4; ++c4) {
            f32x4 v;
            v.x = dX[4 * c4 + 0]; v.y = dX[4 * c4 + 1]; v.z = dX[4 * c4 + 2]; v.w = dX[4 * c4 + 3];
            if (stage) *reinterpret_cast<f32x4*>(s_df + j * SCS + hh * CHH + 4 * c4) = v;
            else *reinterpret_cast<f32x4*>(a.dfeat + (tile_idx * 32 + j) * CH + hh * CHH + 4 * c4) = v;
        }
        float gix = 0.0f, giy = 0.0f, giz = 0.0f;
        if (stage || a.d_pos) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int dz = corner >> 2, dyc = (corner >> 1) & 1, dx = corner & 1;
            const int xi = x0 + dx, yi = y0 + dyc, zi = z0 + dz;
            const bool ok = valid && in_range && xi >= 0 && xi < a.W && yi >= 0 && yi < a.H && zi >= 0 && zi < a.D;
            const float wxc = dx ? wx1 : wx0, wyc = dyc ? wy1 : wy0, wzc = dz ? wz1 : wz0;
            const float w = ok ? __fmul_rn(__fmul_rn(wxc, wyc), wzc) : 0.0f;
            const int xc = min(max(xi, 0), a.W - 1), yc = min(max(yi, 0), a.H - 1), zc = min(max(zi, 0), a.D - 1);
            const long long off = ((long long)(zc * a.H + yc) * a.W + xc) * a.Cs;
            if (stage) { if (hh == 0) s_cw[j * 8 + corner] = w; else s_co[j * 8 + corner] = (int)off; }
            if (a.d_pos && ok) {                           // sampler coordinate gradient (ATen grid_sampler_3d_backward)
                const float* gp = a.grid + off + hh * CHH;
                float dot = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < CHH / 4; ++c4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(gp + 4 * c4);
                    dot = __builtin_fmaf(v.x, dX[4 * c4 + 0], dot); dot = __builtin_fmaf(v.y, dX[4 * c4 + 1], dot);
                    dot = __builtin_fmaf(v.z, dX[4 * c4 + 2], dot); dot = __builtin_fmaf(v.w, dX[4 * c4 + 3], dot);
                }
                gix += (dx ? dot : -dot) * (wyc * wzc);
                giy += (dyc ? dot : -dot) * (wxc * wzc);
                giz += (dz ? dot : -dot) * (wxc * wyc);
            }
        }
        }
        if (stage) {
            __syncthreads();                              // staging visible to every lane that reads it
            const int sp = lane / CH, c = lane % CH;
            if (sp < SPI) {
                for (int i = 0; i < (32 + SPI - 1) / SPI; ++i) {
                    const int smp = i * SPI + sp;
                    if (smp < 32) {
                        const float v = s_df[smp * SCS + c];
#pragma unroll
                        for (int corner = 0; corner < 8; ++corner) {
                            const float w = s_cw[smp * 8 + corner];
                            if (w != 0.0f) atomicAdd(a.d_grid + s_co[smp * 8 + corner] + c, v * w);
                        }
                    }
                }
            }
        }

        LFGC_BSTAMP(4);                                       // geometry + staging + atomic scatter
        // ---- d_pos = direct columns + Fourier embedding + sampler coordinate gradient ---------------------------
        if (a.d_pos) {
            float sk[NF > 0 ? NF : 1][3], ck[NF > 0 ? NF : 1][3];
            bool bad = false;
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const float f = lfgc_freq(k);
                const float a0 = __fmul_rn(p0, f), a1 = __fmul_rn(p1, f), a2 = __fmul_rn(p2, f);
                bad |= lfgc_trig_out_of_range(a0) | lfgc_trig_out_of_range(a1) | lfgc_trig_out_of_range(a2);
                lfgc_sincosf_t<false>(a0, sk[k][0], ck[k][0]);
                lfgc_sincosf_t<false>(a1, sk[k][1], ck[k][1]);
                lfgc_sincosf_t<false>(a2, sk[k][2], ck[k][2]);
            }
            if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const float f = lfgc_freq(k);
                    lfgc_sincosf_t<true>(__fmul_rn(p0, f), sk[k][0], ck[k][0]);
                    lfgc_sincosf_t<true>(__fmul_rn(p1, f), sk[k][1], ck[k][1]);
                    lfgc_sincosf_t<true>(__fmul_rn(p2, f), sk[k][2], ck[k][2]);
                }
            }
            // this lane holds d e[hh*EPH + t] in dX[CHH + t]; evaluate both static mappings, keep this half's
            float dlo[3] = {0.0f, 0.0f, 0.0f}, dhi[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < EPH; ++t) {
                const float d = dX[CHH + t];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int e = half * EPH + t;
                    float* acc3 = half ? dhi : dlo;
                    if (e < 3) {
                        acc3[e] += d;
                    } else if (e < E) {
                        const int k = (e - 3) / 6, wch = (e - 3) % 6;
                        const float f = lfgc_freq(k);
                        if (wch < 3) acc3[wch] += d * f * ck[k][wch];
                        else acc3[wch - 3] -= d * f * sk[k][wch - 3];
                    }
                }
            }
            float g0 = (hh ? dhi[0] : dlo[0]) + gix * (0.5f * (float)a.W);
            float g1 = (hh ? dhi[1] : dlo[1]) + giy * (0.5f * (float)a.H);
            float g2 = (hh ? dhi[2] : dlo[2]) + giz * (0.5f * (float)a.D);
            g0 += __shfl_xor(g0, 32); g1 += __shfl_xor(g1, 32); g2 += __shfl_xor(g2, 32);
            if (valid && hh == 0) {
                a.d_pos[3 * n + 0] = g0; a.d_pos[3 * n + 1] = g1; a.d_pos[3 * n + 2] = g2;
            }
        }
        LFGC_BSTAMP(5);                                       // d_pos
    }
    // the stream's last image (fetched for nobody) must have landed before the workgroup gives its LDS back
    if (NPW > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LFGC_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* dst = a.stamps + ((long long)blockIdx.x * WAVES + wave) * 20;
        for (int k = 0; k < 8; ++k) dst[k] = bst[k];
        dst[16] = __builtin_amdgcn_s_memtime() - bst_t0;
    }
#endif
}

// Deferred scatter of the feature gradients (small batches): d_grid[corner rows of sample n] += w_corner * dfeat[n].
// A wave takes 8 samples, 64 / CH at a time (one atomic wave-instruction = that many whole channel rows), every lane
// forming its sample's corner weights and offsets itself with the forward's arithmetic; thousands of waves, so the
// float atomics' latency (a cold line per row) is covered by occupancy instead of being waited for 128 times in a row.
template <int CH>
__global__ __launch_bounds__(256) void lfgc_bwd_scatter_kernel(const float* __restrict__ pos, const float* __restrict__ dfeat,
                                                              float* __restrict__ d_grid, long long n, int D, int H, int W, int Cs) {
    constexpr int SPI = 64 / CH;                          // samples per wave-instruction
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int sp = lane / CH, c = lane % CH;
    if (sp >= SPI) return;
#pragma unroll 1
    for (int i = 0; i < 8 / SPI + (8 % SPI != 0); ++i) {
        const long long smp = gw * 8 + i * SPI + sp;
        if (i * SPI + sp >= 8 || smp >= n) continue;
        const float p0 = pos[3 * smp], p1 = pos[3 * smp + 1], p2 = pos[3 * smp + 2];
        const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p0, 1.0f), (float)W), 1.0f), 0.5f);
        const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p1, 1.0f), (float)H), 1.0f), 0.5f);
        const float iz = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(p2, 1.0f), (float)D), 1.0f), 0.5f);
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)W);
        const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)H);
        const int z0 = (int)fminf(fmaxf(fz0, -2.0f), (float)D);
        float wx[2], wy[2], wz[2];
        wx[1] = __fsub_rn(ix, fx0); wx[0] = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
        wy[1] = __fsub_rn(iy, fy0); wy[0] = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
        wz[1] = __fsub_rn(iz, fz0); wz[0] = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
        const float v = dfeat[smp * CH + c];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int dz = corner >> 2, dy = (corner >> 1) & 1, dx = corner & 1;
            const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
            const bool ok = (unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H && (unsigned)zi < (unsigned)D;
            const float w = __fmul_rn(__fmul_rn(wx[dx], wy[dy]), wz[dz]);
            if (ok && w != 0.0f) atomicAdd(d_grid + ((long long)(zi * H + yi) * W + xi) * Cs + c, v * w);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------------------
struct LfgcWgradArgs {
    const float* stash;
    const float* dstash;
    const float* d_out;
    long long n;
    long long ntiles;          // 32-sample tiles that hold data (multiple of 4: whole workgroup batches)
    int L;
    float* slabs;              // [gridDim.x / roles][slab_floats]
    int slab_floats;
    int roles;                 // 1: a workgroup computes every layer's gradient for its tiles (one slab per workgroup);
                               // L: workgroup b computes layer b % L only (the last role also the head) for the tiles of
                               // group b / L -- a quarter of the slabs for the same operand reads
    const float* dscale;       // [tiles][L] from the data kernel, or nullptr: exact f32 MFMA contraction
};

// Slab layout (floats): per hidden layer l: dW [HP][NC_l] (NC_0 = K0R in packed column order, else HP) | db [HP];
// then final layer: dWf [HP] | dbf [4].
__host__ __device__ inline int lfgc_slab_layer_off(const LfgcPlan& p, int l) {
    return l == 0 ? 0 : (p.HP * p.K0R + p.HP) + (l - 1) * (p.HP * p.HP + p.HP);
}
__host__ __device__ inline int lfgc_slab_floats(const LfgcPlan& p) { return lfgc_slab_layer_off(p, p.L) + p.HP + 4; }

// The 16 samples a lane contributes to a tile's contraction: [8 kh, 8 kh + 8) and [16 + 8 kh, 16 + 8 kh + 8) of stash row
// `base` (= row start + lane-half offset): exactly the k values lane half kh feeds to the two 16-deep k-steps of
// v_mfma_f32_32x32x16_f16 (and, pairwise with the other half, to 16 v_mfma_f32_32x32x2_f32).
__device__ __forceinline__ void lfgc_load16(const float* __restrict__ base, int kh, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(base + 8 * kh + 16 * (q >> 1) + 4 * (q & 1));
        v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

// One layer's dW (+ db) over this workgroup's sample tiles.  NT = column tiles of this layer, KSIN = stash
// registers of the input (KS0 for layer 0 where the input is the saved x0, else 16*MT pre-activations).
// The workgroup has 8 waves: waves 0-3 and 4-7 run the same tile/column assignment on alternate sample tiles (two
// waves per SIMD, so one wave's loads hide under the other's MFMAs); the second half hands its accumulators over
// through LDS (`s_comb`, [4 waves][TPW*17][64]) and the first half writes the slab.
// Contraction over the 32 samples of a tile: with a.dscale (the f16 builds) as f16 hi/lo split operands, three
// v_mfma_f32_32x32x16_f16 per 16 samples with fp32 accumulation -- dA scaled by the power of two the data kernel chose for
// the tile, the tile's product scaled back exactly before it joins the running sum --, 192 instead of 1024 matrix-pipe
// cycles per 32x32 output tile; a tile whose inputs leave the f16 range (|H| >= 65504: a diverged model), and the exact
// build, take sixteen v_mfma_f32_32x32x2_f32 (bitwise an fp32 fmaf chain).
template <int MT, int NT, bool LAYER0, int KS0>
__device__ __forceinline__ void lfgc_wgrad_layer(const LfgcWgradArgs& a, int l, float* __restrict__ slab_l, int ncol,
                                                 int k0p, long long per_tile, long long dper_tile,
                                                 int lane, int wave8, float* s_comb, int grp, int ngrp) {
    const int half = wave8 >> 2, wave = wave8 & 3;
    constexpr int WPN = 4 / NT;                       // waves sharing one column tile
    constexpr int TPW = (MT + WPN - 1) / WPN;         // row tiles per wave
    const int n_t = wave % NT, msub = wave / NT;
    const int i = lane & 31, kk = lane >> 5;
    f32x16 acc[TPW];
    float dbp[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        dbp[t] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    }
    // B operand row for this lane: column col = 32 n_t + i of the layer input
    const int col = 32 * n_t + i;
    long long boff;
    bool bvalid = true;
    if (LAYER0) {
        const int s = 4 * (col >> 3) + (col & 3), hb = (col >> 2) & 1;
        bvalid = col < k0p;
        boff = (long long)(bvalid ? s : 0) * 64 + hb * 32;
    } else {
        const int r = (i & 3) + 4 * (i >> 3), hb = (i >> 2) & 1;
        boff = 64LL * KS0 + (long long)(l - 1) * (64 * 16 * MT) + (long long)(n_t * 16 + r) * 64 + hb * 32;
    }
    // A operand rows: row = 32 m + i of dA_l
    const int ra = (i & 3) + 4 * (i >> 3), ha = (i >> 2) & 1;
    const long long aoff_base = (long long)l * (64 * 16 * MT) + (long long)ra * 64 + ha * 32;

    for (long long t = grp + (long long)half * ngrp; t < a.ntiles; t += 2LL * ngrp) {
        float Bv[16];
        lfgc_load16(a.stash + t * per_tile + boff, kk, Bv);
        if (LAYER0) {
            if (!bvalid) {
#pragma unroll
                for (int s = 0; s < 16; ++s) Bv[s] = 0.0f;
            }
        } else {
            bool bad = false;
#pragma unroll
            for (int s = 0; s < 16; ++s) bad |= lfgc_trig_out_of_range(Bv[s]);
            float Hv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) Hv[s] = lfgc_snake_t<false>(Bv[s]);
            if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
                for (int s = 0; s < 16; ++s) Hv[s] = lfgc_snake_t<true>(Bv[s]);
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) Bv[s] = Hv[s];
        }
        bool split = a.dscale != nullptr;
        float sc = 1.0f, isc = 1.0f;
        h16x8 Bhi[2], Blo[2];
        if (split) {
            float bmax = 0.0f;
#pragma unroll
            for (int s = 0; s < 16; s += 2) bmax = lfgc_absmax3(bmax, Bv[s], Bv[s + 1]);
            split = !__any(!(bmax < 65504.0f));                    // wave-uniform; NaN / inf / huge inputs -> exact path
            sc = a.dscale[t * a.L + l];
            isc = __int_as_float(0x7F000000 - __float_as_int(sc));  // 1 / 2^k, exact
            lfgc_split8(Bv, Bhi[0], Blo[0]);
            lfgc_split8(Bv + 8, Bhi[1], Blo[1]);
        }
#pragma unroll
        for (int tw = 0; tw < TPW; ++tw) {
            const int m = msub + tw * WPN;
            if (m < MT) {
                float Av[16];
                lfgc_load16(a.dstash + t * dper_tile + aoff_base + (long long)m * (16 * 64), kk, Av);
#pragma unroll
                for (int s = 0; s < 16; ++s) dbp[tw] += Av[s];
                if (split) {
                    float As[16];
#pragma unroll
                    for (int s = 0; s < 16; ++s) As[s] = Av[s] * sc;
                    h16x8 Ahi[2], Alo[2];
                    lfgc_split8(As, Ahi[0], Alo[0]);
                    lfgc_split8(As + 8, Ahi[1], Alo[1]);
                    f32x16 tmp;
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmp[r] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[ks], Bhi[ks], tmp, 0, 0, 0);
                        tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[ks], Blo[ks], tmp, 0, 0, 0);
                        tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[ks], Bhi[ks], tmp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tw][r] = __builtin_fmaf(tmp[r], isc, acc[tw][r]);
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        acc[tw] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[s], Bv[s], acc[tw], 0, 0, 0);
                }
            }
        }
    }
    // second half -> LDS -> first half
    float* mine = s_comb + (long long)wave * (TPW * 17 * 64) + lane;
    if (half == 1) {
#pragma unroll
        for (int tw = 0; tw < TPW; ++tw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(tw * 17 + r) * 64] = acc[tw][r];
            mine[(tw * 17 + 16) * 64] = dbp[tw];
        }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int tw = 0; tw < TPW; ++tw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tw][r] += mine[(tw * 17 + r) * 64];
            dbp[tw] += mine[(tw * 17 + 16) * 64];
        }
    }
    __syncthreads();
    if (half == 1) return;
    // write this workgroup's partial: dW[32m + row][32 n_t + colj], rows from the accumulator layout
    const int cj = lane & 31, hc = lane >> 5;
#pragma unroll
    for (int tw = 0; tw < TPW; ++tw) {
        const int m = msub + tw * WPN;
        if (m < MT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hc;
                slab_l[(long long)row * ncol + 32 * n_t + cj] = acc[tw][r];
            }
            const float tot = dbp[tw] + __shfl_xor(dbp[tw], 32);
            if (n_t == 0 && kk == 0) slab_l[(long long)(32 * MT) * ncol + 32 * m + i] = tot;
        }
    }
}

template <int CH, int MT, int NF>
__global__ __launch_bounds__(512, 2) void lfgc_bwd_weight_kernel(const LfgcWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_comb[];
    constexpr int E = 3 + 6 * NF;
    constexpr int EP = (E + 7) / 8 * 8;
    constexpr int K0P = CH + EP;
    constexpr int K0R = (K0P + 31) / 32 * 32;
    constexpr int KS0 = K0P / 2;
    constexpr int HP = 32 * MT;
    constexpr int NT0 = K0R / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L;
    const long long per_tile = 64LL * (KS0 + L * 16 * MT);
    const long long dper_tile = 64LL * (L * 16 * MT);
    const int R = a.roles;
    const int role = (int)blockIdx.x % R, grp = (int)blockIdx.x / R, ngrp = (int)gridDim.x / R;    // workgroup-uniform
    float* slab = a.slabs + (long long)grp * a.slab_floats;

    if (R == 1 || role == 0) lfgc_wgrad_layer<MT, NT0, true, KS0>(a, 0, slab, K0R, K0P, per_tile, dper_tile, lane, wave, s_comb, grp, ngrp);
    for (int l = 1; l < L; ++l) {
        if (R != 1 && role != l) continue;
        float* slab_l = slab + (HP * K0R + HP) + (long long)(l - 1) * (HP * HP + HP);
        lfgc_wgrad_layer<MT, MT, false, KS0>(a, l, slab_l, HP, K0P, per_tile, dper_tile, lane, wave, s_comb, grp, ngrp);
    }

    // final Linear: dWf[k] = sum_n dy_n H_L[n,k], dbf = sum_n dy_n.  Wave w owns column tile w.
    if (R == 1 || role == L - 1) {
        float* slab_f = slab + (HP * K0R + HP) + (long long)(L - 1) * (HP * HP + HP);
        const int i = lane & 31, kk = lane >> 5;
        const int half = wave >> 2, w4 = wave & 3;
        float wsum = 0.0f, bsum = 0.0f;
        if (w4 < MT) {
            const int r = (i & 3) + 4 * (i >> 3), hb = (i >> 2) & 1;
            const long long boff = 64LL * KS0 + (long long)(L - 1) * (64 * 16 * MT) + (long long)(w4 * 16 + r) * 64 + hb * 32;
            for (long long t = grp + (long long)half * ngrp; t < a.ntiles; t += 2LL * ngrp) {
                float Bv[16];
                lfgc_load16(a.stash + t * per_tile + boff, kk, Bv);
                bool bad = false;
#pragma unroll
                for (int s = 0; s < 16; ++s) bad |= lfgc_trig_out_of_range(Bv[s]);
                float Hv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) Hv[s] = lfgc_snake_t<false>(Bv[s]);
                if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) Hv[s] = lfgc_snake_t<true>(Bv[s]);
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const long long smp = t * 32 + 8 * kk + 16 * (s >> 3) + (s & 7);       // lfgc_load16's sample order
                    const float dyv = smp < a.n ? a.d_out[smp] : 0.0f;
                    wsum = __builtin_fmaf(dyv, Hv[s], wsum);
                    bsum += dyv;
                }
            }
            wsum += __shfl_xor(wsum, 32);
            bsum += __shfl_xor(bsum, 32);
        }
        if (half == 1) { s_comb[w4 * 128 + lane] = wsum; s_comb[w4 * 128 + 64 + lane] = bsum; }
        __syncthreads();
        if (half == 0 && w4 < MT) {
            wsum += s_comb[w4 * 128 + lane];
            bsum += s_comb[w4 * 128 + 64 + lane];
            if (kk == 0) slab_f[32 * w4 + i] = wsum;
            if (w4 == 0 && lane == 0) slab_f[HP] = bsum;
        }
    }
}

struct LfgcReduceArgs {
    const float* slabs;
    int nslabs, slab_floats;
    float* dw[LFGC_MAX_LAYERS + 1];
    float* db[LFGC_MAX_LAYERS + 1];
    LfgcPlan plan;
    int col_of_src[64];        // layer 0: packed column that holds original column c
};

// d_weights / d_biases in nn.Linear layout = sum over workgroup slabs.  A block owns 64 consecutive output
// elements; its 4 waves each sum a quarter of the slabs (coalesced 256-B rows), the quarters are combined in a
// fixed order through LDS, so the result is bitwise repeatable.
static __global__ __launch_bounds__(256) void lfgc_bwd_reduce_kernel(const LfgcReduceArgs a) {
    const LfgcPlan& p = a.plan;
    const int K0 = p.E + p.C;
    const int n0 = p.H * K0 + p.H;                     // layer 0: weights then bias
    const int n1 = p.H * p.H + p.H;
    const int total = n0 + (p.L - 1) * n1 + p.H + 1;
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    int src = 0;
    float* dst = nullptr;
    if (idx < total) {
        if (idx < n0) {
            if (idx < p.H * K0) {
                const int row = idx / K0, c = idx % K0;
                src = row * p.K0R + a.col_of_src[c];
                dst = a.dw[0] + idx;
            } else {
                const int r = idx - p.H * K0;
                src = p.HP * p.K0R + r;
                dst = a.db[0] + r;
            }
        } else if (idx < n0 + (p.L - 1) * n1) {
            const int l = 1 + (idx - n0) / n1, oo = (idx - n0) % n1;
            const int base = lfgc_slab_layer_off(p, l);
            if (oo < p.H * p.H) {
                src = base + (oo / p.H) * p.HP + (oo % p.H);
                dst = a.dw[l] + oo;
            } else {
                src = base + p.HP * p.HP + (oo - p.H * p.H);
                dst = a.db[l] + (oo - p.H * p.H);
            }
        } else {
            const int oo = idx - n0 - (p.L - 1) * n1;
            const int base = lfgc_slab_layer_off(p, p.L);
            if (oo < p.H) { src = base + oo; dst = a.dw[p.L] + oo; }
            else { src = base + p.HP; dst = a.db[p.L]; }
        }
    }
    float s = 0.0f;
    if (idx < total) {
        const float* sp = a.slabs + src;
        for (int g = q; g < a.nslabs; g += 4) s += sp[(long long)g * a.slab_floats];
    }
    part[q][o] = s;
    __syncthreads();
    if (q == 0 && idx < total) *dst = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
}

template <int CH, int MT, int NF, int WAVES, int PREC>
static int lfgc_launch_bwd_data(const LfgcBwdArgs& a, int lds_bytes, int grid_data, hipStream_t stream) {
    auto kd = lfgc_bwd_data_kernel<CH, MT, NF, WAVES, PREC>;
    static int lds_limit_set[LFGC_MAX_DEVICES] = {0};      // per (instantiation, device)
    const int dev = lfgc_current_device();
    if (lds_bytes > 64 * 1024 && lds_bytes > lds_limit_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kd),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        lds_limit_set[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(kd, dim3(grid_data), dim3(WAVES * 64), lds_bytes, stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

// waves = waves per workgroup of the data kernel (4 or 8, chosen by the caller together with a.nbatches)
template <int CH, int MT, int NF>
static int lfgc_launch_bwd(const LfgcBwdArgs& a, const LfgcWgradArgs& w, int waves, int h16, int lds_bytes, int grid_data,
                           int grid_w, hipStream_t stream) {
    int rc;
    // h16 = the C-ABI precision code: 0 exact f32 MFMA chain, 1 f16 hi/lo split, 2 single f16 product
    if (h16 == 1) rc = waves == 8 ? lfgc_launch_bwd_data<CH, MT, NF, 8, 1>(a, lds_bytes, grid_data, stream)
                                  : lfgc_launch_bwd_data<CH, MT, NF, 4, 1>(a, lds_bytes, grid_data, stream);
    else if (h16 == 2) rc = waves == 8 ? lfgc_launch_bwd_data<CH, MT, NF, 8, 2>(a, lds_bytes, grid_data, stream)
                                       : lfgc_launch_bwd_data<CH, MT, NF, 4, 2>(a, lds_bytes, grid_data, stream);
    else rc = waves == 8 ? lfgc_launch_bwd_data<CH, MT, NF, 8, 0>(a, lds_bytes, grid_data, stream)
                         : lfgc_launch_bwd_data<CH, MT, NF, 4, 0>(a, lds_bytes, grid_data, stream);
    if (rc != LFGC_OK) return rc;
    if (a.dfeat) {                                        // deferred scatter of the feature gradients
        const long long blocks = (a.n + 31) / 32;         // 4 waves x 8 samples
        hipLaunchKernelGGL(lfgc_bwd_scatter_kernel<CH>, dim3((unsigned)blocks), dim3(256), 0, stream, a.pos, a.dfeat, a.d_grid, a.n,
                           a.D, a.H, a.W, a.Cs);
        LFGC_HIP_CHECK_LAUNCH();
    }
    {
        constexpr int K0R_ = (CH + (3 + 6 * NF + 7) / 8 * 8 + 31) / 32 * 32;
        constexpr int NT0_ = K0R_ / 32;
        constexpr int TPW0 = (MT + 4 / NT0_ - 1) / (4 / NT0_), TPW1 = (MT + 4 / MT - 1) / (4 / MT);
        constexpr int TPWM = TPW0 > TPW1 ? TPW0 : TPW1;
        const int comb_bytes = 4 * TPWM * 17 * 64 * 4;
        auto kw = lfgc_bwd_weight_kernel<CH, MT, NF>;
        static int comb_limit_set[LFGC_MAX_DEVICES] = {0};
        const int dev = lfgc_current_device();
        if (comb_bytes > 64 * 1024 && comb_bytes > comb_limit_set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kw),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, comb_bytes);
            if (e != hipSuccess) return (int)e;
            comb_limit_set[dev] = comb_bytes;
        }
        hipLaunchKernelGGL(kw, dim3(grid_w), dim3(512), comb_bytes, stream, w);
    }
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}
