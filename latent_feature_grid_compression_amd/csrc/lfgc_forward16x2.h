// lfgc_forward16x2.h -- the f16-split forward with TWO 32-sample tiles per wave and ONE wave per SIMD (4-wave workgroups,
// one per CU, up to 512 registers per lane): lattice mode (z-run tiles + column sampler), streamed nets, no stash.
//
// Same arithmetic, images, ring, epilogue and weight stream as lfgc_forward16.h -- every function here is the two-tile
// form of the one of the same name there.  What changes is the instruction stream of a wave: every A-operand fragment
// read from LDS feeds the MFMAs of both tiles (half the ds_read_b128 per sample), two MFMAs on independent accumulators
// go out back to back, and the two tiles' epilogue slices alternate in the gaps, so that the in-order wave always has
// independent work next -- what the second wave of a SIMD provides in the one-tile kernel.  A workgroup still covers 8
// tiles = 256 samples per pass over the weight stream.
#pragma once
#include "lfgc_forward16.h"

// The gaps of one output tile for both sample tiles T = 0, 1 (see lfgc_tile_gaps): per (k-step, product) two MFMAs
// sharing the A operand, then that gap's slice of the weight stream, the next operand read, and the slices of the two
// pending epilogues.
template <int KS16, bool SPLIT, int GA, bool E_IS_IN_TAIL, int NP, int NVEC, int DG0, int DSPAN, int WAVES, class EPI>
__device__ __forceinline__ void lfgc_tile_gaps2(const float* __restrict__ arow, const float* __restrict__ arow_next,
                                                u32x4 (&INhi)[2][KS16], u32x4 (&INlo)[2][KS16], f32x16 (&acc)[2],
                                                LfgcOperands& w, EPI (&ep)[2], const f32x16 (&eacc)[2],
                                                u32x4 (&Ehi)[2][2], u32x4 (&Elo)[2][2], float (&ydot)[2], float (&tmax)[2],
                                                const LfgcDmaPlan& dma) {
    constexpr int MPK = SPLIT ? 3 : 1;
    static_assert(!E_IS_IN_TAIL || GA <= (KS16 - 2) * MPK, "a carried tile must be done before the k-steps that read it");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    lfgc_static_for<KS16>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (E_IS_IN_TAIL && ks == KS16 - 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                INhi[t][KS16 - 2] = Ehi[t][0]; INhi[t][KS16 - 1] = Ehi[t][1];
                INlo[t][KS16 - 2] = Elo[t][0]; INlo[t][KS16 - 1] = Elo[t][1];
            }
        }
        const h16x8 whi = w.hi[0], wlo = w.lo[0];          // this k-step's operands (read LFGC_PF k-steps ago)
        h16x8 nhi = w.hi[LFGC_PF - 1], nlo = w.lo[LFGC_PF - 1];
        lfgc_static_for<MPK>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            constexpr int g = ks * MPK + u;
            // Each of the two MFMAs gets its own gap: a second MFMA issued right behind the first waits for the matrix
            // pipe (32 cycles) with the wave's vector issue blocked behind it -- the first one's shadow is lost (measured:
            // ~100 cycles per pair that way).  Tile 0's gap carries the stream piece, the operand read and tile 0's epilogue
            // slice, tile 1's gap tile 1's slice.
            lfgc_static_for<2>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                const h16x8 bh = __builtin_bit_cast(h16x8, INhi[t][ks]);
                if (SPLIT) {
                    const h16x8 bl = __builtin_bit_cast(h16x8, INlo[t][ks]);
                    if (u == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh, acc[t], 0, 0, 0);       // small terms first
                    if (u == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bl, acc[t], 0, 0, 0);
                    if (u == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh, acc[t], 0, 0, 0);
                } else {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh, acc[t], 0, 0, 0);
                }
                if constexpr (t == 0) {
                    if constexpr (NP > 0) {          // this gap's share of the weight stream, in the MFMA's shadow
                        constexpr int gg = DG0 + g;
                        constexpr int pi_lo = gg < DSPAN ? (gg * NP + DSPAN - 1) / DSPAN : NP;
                        constexpr int pi_hi = gg + 1 < DSPAN ? ((gg + 1) * NP + DSPAN - 1) / DSPAN : NP;
                        lfgc_static_for<(gg < DSPAN ? pi_hi - pi_lo : 0)>([&](auto p_c) { lfgc_dma_piece_ct<WAVES, pi_lo + decltype(p_c)::value, NVEC>(dma); });
                    }
                    const float* nsrc = (ks + LFGC_PF < KS16) ? arow + 16 * (ks + LFGC_PF)
                                                              : (arow_next ? arow_next + 16 * (ks + LFGC_PF - KS16) : nullptr);
                    if (nsrc) {
                        if (u == 0) nhi = *reinterpret_cast<const h16x8*>(nsrc);
                        if (SPLIT && u == 1) nlo = *reinterpret_cast<const h16x8*>(nsrc + 4);
                    }
                }
                if constexpr (GA > 0 && g < GA) lfgc_epilogue_gap<GA, g>(ep[t], eacc[t], Ehi[t], Elo[t], ydot[t], tmax[t]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#pragma unroll
        for (int d = 0; d + 1 < LFGC_PF; ++d) { w.hi[d] = w.hi[d + 1]; w.lo[d] = w.lo[d + 1]; }
        w.hi[LFGC_PF - 1] = nhi; w.lo[LFGC_PF - 1] = nlo;
    });
}

// One layer on the wave's two tiles (see lfgc_layer_fwd16; never with a stash).
template <int KS16, int MT, int S, bool LAST, bool SPLIT, bool HAS_CARRY, int NP, int NVEC, int WAVES>
__device__ __forceinline__ void lfgc_layer_fwd16x2(const float* __restrict__ s_blk, u32x4 (&INhi)[2][KS16], u32x4 (&INlo)[2][KS16],
                                                   const LfgcCarry (&carry)[2], float inv_scale, const float* __restrict__ s_bias,
                                                   u32x4 (&OUThi)[2][2 * MT], u32x4 (&OUTlo)[2][2 * MT], LfgcCarry (&out_carry)[2],
                                                   const float* __restrict__ s_final, float (&ydot)[2], float (&tmax)[2],
                                                   int j, int hh, const LfgcDmaPlan& dma) {
    constexpr int MPK = SPLIT ? 3 : 1;
    constexpr int G = KS16 * MPK;
    constexpr int DSPAN = (MT * G * LFGC_DMA_SPAN4 + 3) / 4;
    const float* s_row = s_blk + j * S + 8 * hh;
    const float* bias_l = s_bias + 4 * hh;
    const float* wf_l = s_final + 4 * hh;
    LfgcOperands w;
#pragma unroll
    for (int d = 0; d < LFGC_PF; ++d) {
        w.hi[d] = *reinterpret_cast<const h16x8*>(s_row + 16 * d);
        w.lo[d] = w.hi[d];
        if (SPLIT) w.lo[d] = *reinterpret_cast<const h16x8*>(s_row + 16 * d + 4);
    }

    f32x16 accs[2][2];            // [ping-pong][tile]
    {   // output tile 0: shadows the previous layer's last output tile, which produces this layer's last two input fragments
        LfgcEpilogue<false, false, SPLIT> ep[2];
        u32x4 ehi[2][2], elo[2][2];
        constexpr int GA0 = HAS_CARRY ? (KS16 - 2) * MPK : 0;
        f32x16 cacc[2];
        if (HAS_CARRY) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ep[t].inv_scale = carry[t].inv_scale; ep[t].bias = carry[t].bias; ep[t].wf = nullptr; ep[t].stash = nullptr;
                cacc[t] = carry[t].acc;
                if constexpr (GA0 == 0) {
                    ep[t].begin();
                    ep[t].all(carry[t].acc, ehi[t], elo[t], ydot[t], tmax[t]);
                    INhi[t][KS16 - 2] = ehi[t][0]; INhi[t][KS16 - 1] = ehi[t][1]; INlo[t][KS16 - 2] = elo[t][0]; INlo[t][KS16 - 1] = elo[t][1];
                }
            }
        }
        lfgc_tile_gaps2<KS16, SPLIT, GA0, (HAS_CARRY && GA0 > 0), NP, NVEC, 0, DSPAN, WAVES>(s_row, MT > 1 ? s_row + 32 * S : nullptr, INhi, INlo,
                                                    accs[0], w, ep, cacc, ehi, elo, ydot, tmax, dma);
    }
    lfgc_static_for<MT - 1>([&](auto m_c) {
        constexpr int m = 1 + decltype(m_c)::value;
        LfgcEpilogue<false, LAST, SPLIT> ep[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ep[t].inv_scale = inv_scale; ep[t].bias = bias_l + 32 * (m - 1); ep[t].wf = wf_l + 32 * (m - 1); ep[t].stash = nullptr;
        }
        u32x4 ehi[2][2], elo[2][2];
        lfgc_tile_gaps2<KS16, SPLIT, G, false, NP, NVEC, m * G, DSPAN, WAVES>(s_row + 32 * m * S, m + 1 < MT ? s_row + 32 * (m + 1) * S : nullptr,
                                              INhi, INlo, accs[m & 1], w, ep, accs[(m - 1) & 1], ehi, elo, ydot, tmax, dma);
        if (!LAST) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                OUThi[t][2 * (m - 1)] = ehi[t][0]; OUThi[t][2 * (m - 1) + 1] = ehi[t][1];
                OUTlo[t][2 * (m - 1)] = elo[t][0]; OUTlo[t][2 * (m - 1) + 1] = elo[t][1];
            }
        }
    });
    if (LAST) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            LfgcEpilogue<false, true, SPLIT> ep;
            ep.inv_scale = inv_scale; ep.bias = bias_l + 32 * (MT - 1); ep.wf = wf_l + 32 * (MT - 1); ep.stash = nullptr;
            u32x4 ehi[2], elo[2];
            ep.begin();
            ep.all(accs[(MT - 1) & 1][t], ehi, elo, ydot[t], tmax[t]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            out_carry[t].acc = accs[(MT - 1) & 1][t];
            out_carry[t].inv_scale = inv_scale;
            out_carry[t].bias = bias_l + 32 * (MT - 1);
            out_carry[t].stash = nullptr;
        }
    }
}

// Lattice mode only (LfgcFwdArgs::zrun), streamed nets, 4 waves, one workgroup per CU.  nbatches counts passes of 8 tiles:
// wave w of pass b takes tiles 8 b + 2 w and 8 b + 2 w + 1.
template <int CH, int MT, int NF, bool SPLIT>
__global__ __launch_bounds__(256, 1) void lfgc_fwd16x2_kernel(const LfgcFwdArgs a) {
    constexpr int WAVES = 4;
    constexpr int E = 3 + 6 * NF;
    constexpr int EP = (E + 7) / 8 * 8;
    constexpr int K0P = CH + EP;
    constexpr int KS0 = K0P / 2;
    constexpr int K0P16 = (K0P + 15) / 16 * 16;
    constexpr int KS16_0 = K0P16 / 16;
    constexpr int HP = 32 * MT;
    constexpr int KS16_1 = HP / 16;
    constexpr int S0 = K0P16 + 4;
    constexpr int S1 = HP + 4;
    constexpr int BLK0 = HP * S0 + HP;
    constexpr int BLK1 = HP * S1 + HP;
    constexpr int BLKMAX = BLK0 > BLK1 ? BLK0 : BLK1;
    constexpr int NT = WAVES * 64;
    constexpr int NP1 = ((BLK1 / 4 + 63) / 64 + WAVES - 1) / WAVES;
    constexpr int NP0 = ((BLK0 / 4 + 63) / 64 + WAVES - 1) / WAVES;
    constexpr int F_BLK0 = HP * (K0P + 4) + HP, F_BLK1 = HP * (HP + 4) + HP;
    constexpr int K0R = (K0P + 31) / 32 * 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_final = smem;
    float* s_scale = smem + HP + 4;
    float* s_bias = s_scale + 16;
    float* s_w = s_bias + LFGC_MAX_LAYERS * HP;
    float* s_coord = s_w + 2 * BLKMAX;
    float* s_colbase = s_coord + ((a.res0 + a.res1 + a.res2 + 3) & ~3);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31;
    const int hh = lane >> 5;
    const int L = a.L;
    const int off_final = F_BLK0 + (L - 1) * F_BLK1;
    const int off_h = off_final + HP + 4 + K0R * (HP + 4) + (L - 1) * HP * (HP + 4);
    const float* hblk = a.packed + off_h + 32 + LFGC_MAX_LAYERS * HP + HP;
    float* s_col[2] = {s_colbase + (2 * wave) * (a.nzc * (CH + 4)), s_colbase + (2 * wave + 1) * (a.nzc * (CH + 4))};

    for (int i = tid; i < HP; i += NT) s_final[i] = a.packed[off_h + 32 + LFGC_MAX_LAYERS * HP + i];
    if (tid < 4) s_final[HP + tid] = a.packed[off_final + HP + tid];
    if (tid < 16) s_scale[tid] = a.packed[off_h + 16 + tid];
    for (int i = tid; i < L * HP; i += NT) s_bias[i] = a.packed[off_h + 32 + i];
    lfgc_dma_to_lds(hblk, s_w, BLK0, wave, lane, WAVES);
    {
        const int r01 = a.res0 + a.res1, r012 = r01 + a.res2;
        for (int i = tid; i < r012; i += NT) {
            s_coord[i] = i < a.res0 ? lfgc_lattice_coord(i, a.res0, a.tile, a.scale0)
                       : i < r01 ? lfgc_lattice_coord(i - a.res0, a.res1, a.tile, a.scale1)
                                 : lfgc_lattice_coord(i - r01, a.res2, a.tile, a.scale2);
        }
    }
    __syncthreads();
    unsigned step = 0;
#ifdef LFGC_STAMPS
    unsigned long long st_acc[16] = {0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // the wave's tile pair walks the z-run tiles of the slab: pair index p = 4 batch + wave -> tiles 2 p, 2 p + 1; both walks
    // kept in wave-uniform (x, y, run) form and advanced by the grid stride with carries (lfgc_forward16.h)
    int zx[2], zy[2], zt[2], dzx, dzy, dzt;
    {
        const unsigned tpr = (unsigned)a.tiles_per_row, r1 = (unsigned)a.res1;
        const unsigned stride = gridDim.x * (2 * WAVES), drow = stride / tpr;
        dzt = (int)(stride - drow * tpr); dzx = (int)(drow / r1); dzy = (int)(drow - (drow / r1) * r1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(2 * (blockIdx.x * WAVES + wave) + t));
            const unsigned row0 = t0 / tpr;
            zt[t] = (int)(t0 - row0 * tpr); zx[t] = (int)(row0 / r1); zy[t] = (int)(row0 - (row0 / r1) * r1);
        }
    }

    for (long long batch = blockIdx.x; batch < a.nbatches; batch += gridDim.x) {
        long long n[2];
        bool valid[2];
        LfgcColumnSampler<CH, NF> cs[2];
        LFGC_STAMP(0);            // loop overhead / previous tail
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long tile_idx = 2 * (batch * WAVES + wave) + t;
            const bool tile_ok = tile_idx < a.ntiles;
            const int vx = tile_ok ? zx[t] : 0, vy = tile_ok ? zy[t] : 0, vz = (tile_ok ? zt[t] : 0) * LFGC_TILE_SAMPLES + j;
            valid[t] = tile_ok && vz < a.res2;
            n[t] = ((long long)vx * a.res1 + vy) * a.res2 + vz;
            cs[t].stage_a(a, a.x_begin + vx, vy, min(vz, a.res2 - 1), s_coord, s_col[t], lane);
            zt[t] += dzt;
            const int c1 = zt[t] >= a.tiles_per_row ? 1 : 0;
            zt[t] -= c1 * a.tiles_per_row;
            zy[t] += dzy + c1;
            const int c2 = zy[t] >= a.res1 ? 1 : 0;
            zy[t] -= c2 * a.res1;
            zx[t] += dzx + c2;
        }

        LfgcDmaPlan dma = {hblk, s_w, BLK0 / 4, wave_s, (unsigned)lane * 16u, 0ull, 0u};
        asm volatile("" : "+s"(dma.wave));
        auto acquire = [&](int l) -> const float* {
            LFGC_STAMP(2 + 2 * (l < 6 ? l : 6));
            // my pieces of this layer's block have landed; for layer 0 the column loads (4 per pass and tile) stay in flight
            if (l == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * 4 * LfgcColumnSampler<CH, NF>::NPASS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LFGC_STAMP(1);
            __syncthreads();
            LFGC_STAMP(3 + 2 * (l < 5 ? l : 5));
            const float* blk = s_w + (step & 1) * BLKMAX;
            const int ln = (l + 1 == L) ? 0 : l + 1;
            dma.src = hblk + (ln == 0 ? 0 : BLK0 + (long long)(ln - 1) * BLK1);
            dma.dst = s_w + ((step + 1) & 1) * BLKMAX;
            dma.nvec = (ln == 0 ? BLK0 : BLK1) / 4;
            lfgc_dma_plan_block(dma);
            ++step;
            return blk;
        };

        float ydot[2] = {0.0f, 0.0f}, tmax[2] = {0.0f, 0.0f};
        u32x4 Ahi[2][2 * MT], Alo[2][2 * MT], Bhi[2][2 * MT], Blo[2][2 * MT];
        LfgcCarry ca[2], cb[2];
        {   // layer 0
            const float* blk = acquire(0);
            u32x4 X0hi[2][KS16_0], X0lo[2][KS16_0];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float X[8 * KS16_0];
                {
                    float B0[KS0];
                    cs[t].stage_b(s_col[t], hh, B0);
#pragma unroll
                    for (int s = 0; s < KS0; ++s) X[s] = B0[s];
#pragma unroll
                    for (int s = KS0; s < 8 * KS16_0; ++s) X[s] = 0.0f;
                }
#pragma unroll
                for (int s = 0; s < KS16_0; ++s) {
                    h16x8 fh, fl;
                    if (SPLIT) lfgc_split8(X + 8 * s, fh, fl);
                    else { lfgc_cvt8(X + 8 * s, fh); fl = fh; }
                    X0hi[t][s] = __builtin_bit_cast(u32x4, fh); X0lo[t][s] = __builtin_bit_cast(u32x4, fl);
                }
            }
            if (L == 1)
                lfgc_layer_fwd16x2<KS16_0, MT, S0, true, SPLIT, false, NP0, BLK0 / 4, WAVES>(blk, X0hi, X0lo, ca, s_scale[8], s_bias, Ahi, Alo, ca,
                                                                                 s_final, ydot, tmax, j, hh, dma);
            else
                lfgc_layer_fwd16x2<KS16_0, MT, S0, false, SPLIT, false, NP1, BLK1 / 4, WAVES>(blk, X0hi, X0lo, ca, s_scale[8], s_bias, Ahi, Alo, ca,
                                                                                  s_final, ydot, tmax, j, hh, dma);
        }
        {
            int l = 1;
            for (; l + 2 < L; l += 2) {
                const float* blk = acquire(l);
                lfgc_layer_fwd16x2<KS16_1, MT, S1, false, SPLIT, true, NP1, BLK1 / 4, WAVES>(blk, Ahi, Alo, ca, s_scale[8 + l], s_bias + l * HP, Bhi, Blo, cb,
                                                                                 s_final, ydot, tmax, j, hh, dma);
                blk = acquire(l + 1);
                lfgc_layer_fwd16x2<KS16_1, MT, S1, false, SPLIT, true, NP1, BLK1 / 4, WAVES>(blk, Bhi, Blo, cb, s_scale[9 + l], s_bias + (l + 1) * HP, Ahi, Alo, ca,
                                                                                 s_final, ydot, tmax, j, hh, dma);
            }
            if (l + 1 < L) {
                const float* blk = acquire(l);
                lfgc_layer_fwd16x2<KS16_1, MT, S1, false, SPLIT, true, NP1, BLK1 / 4, WAVES>(blk, Ahi, Alo, ca, s_scale[8 + l], s_bias + l * HP, Bhi, Blo, cb,
                                                                                 s_final, ydot, tmax, j, hh, dma);
                ++l;
                blk = acquire(l);
                lfgc_layer_fwd16x2<KS16_1, MT, S1, true, SPLIT, true, NP0, BLK0 / 4, WAVES>(blk, Bhi, Blo, cb, s_scale[8 + l], s_bias + l * HP, Ahi, Alo, ca,
                                                                                s_final, ydot, tmax, j, hh, dma);
            } else if (l < L) {
                const float* blk = acquire(l);
                lfgc_layer_fwd16x2<KS16_1, MT, S1, true, SPLIT, true, NP0, BLK0 / 4, WAVES>(blk, Ahi, Alo, ca, s_scale[8 + l], s_bias + l * HP, Bhi, Blo, cb,
                                                                                s_final, ydot, tmax, j, hh, dma);
            }
        }
        LFGC_STAMP(14);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y = ydot[t] + __shfl_xor(ydot[t], 32);
            y += s_final[HP];
            float tm = fmaxf(tmax[t], __shfl_xor(tmax[t], 32));
            if (!(tm <= LFGC_TURNS_MAX)) y = __builtin_nanf("");
            if (a.status && valid[t] && !(__builtin_fabsf(y) < __builtin_inff()))
                __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.clamp) y = fminf(fmaxf(y, -1.0f), 1.0f);
            if (valid[t] && hh == 0) a.out[n[t]] = y;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stream's last block (fetched for nobody)
#ifdef LFGC_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* dst = a.stamps + ((long long)blockIdx.x * 8 + wave) * 20;
        for (int k = 0; k < 16; ++k) dst[k] = st_acc[k];
        dst[16] = __builtin_amdgcn_s_memtime() - st_t0;
        dst[17] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
}

template <int CH, int MT, int NF>
static int lfgc_launch_fwd16x2(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    auto launch = [&](auto kern) -> int {
        static int lds_limit_set[2][LFGC_MAX_DEVICES] = {{0}};
        const int dev = lfgc_current_device(), which = a.single ? 1 : 0;
        if (lds_bytes > 64 * 1024 && lds_bytes > lds_limit_set[which][dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return (int)e;
            lds_limit_set[which][dev] = lds_bytes;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, stream, a);
        LFGC_HIP_CHECK_LAUNCH();
        return LFGC_OK;
    };
    return a.single ? launch(lfgc_fwd16x2_kernel<CH, MT, NF, false>) : launch(lfgc_fwd16x2_kernel<CH, MT, NF, true>);
}
