// lfgc_forward16.h -- the fused forward with the MLP contractions on the f16 matrix pipe at fp32-level accuracy.
//
// Same boundary, mapping and data flow as lfgc_forward.h (replaces model/Feature_Grid_Model.py:62-78), but every
// fp32 operand of the layer GEMMs is carried as an f16 pair x = hi + lo (hi = rne_f16(x), lo = rne_f16(x - hi):
// 22-24 significant bits) and each fp32 product block is three v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//        W.h  ~=  W_hi.h_hi + W_hi.h_lo + W_lo.h_hi            (the dropped W_lo.h_lo term is <= 2^-22 relative)
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the f16 rate (64 cycles per 4 kFLOP vs 32 cycles per 32 kFLOP), so
// three f16 MFMAs do the work of eight f32 ones in 96 instead of 512 pipe cycles.
//
// Register mapping: one wave = 32 samples; D/C layout of the 32x32x16 MFMA equals the 32x32x2 one, so the chain
// "accumulators -> SnakeAlt -> next layer's B operand" is kept: accumulator registers 8s..8s+7 of tile m become, as
// an f16x8 fragment, exactly the B operand of k-step 2m + s (the weight columns are stored in that k order).
//
// Pre-activations in TURNS OF PI.  The f16-split weight images and the biases of every hidden layer are stored
// divided by pi (pack_kernel, in fp64 before the hi/lo split: no precision is lost against the split itself), so the
// accumulator holds t = a / pi and
//     SnakeAlt(a) = a/2 + sin(a)^2 = (pi/2) t + (1 - cos(2 pi t))/2,      cos(2 pi t) = v_cos_f32(t)
// (the hardware cosine takes revolutions and reduces its argument exactly; its absolute error is 1.25e-7,
// tools/microbench/hwcos_snake.hip): 4 VALU instructions per activation instead of 13 for Cody-Waite + polynomial.
// v_cos_f32 is only valid for |t| <= 256: the activations handed to the next layer are multiplied by LFGC_ACT_SCALE
// (folded into the constants: free; the next layer's weight image is divided by it), chosen so that their f16 hi
// half overflows to inf before |t| reaches 256 (lfgc_common.h).  Such an overflow -- as any |grid feature| >= 65520 --
// turns the sample's output into NaN; the kernel reports it through the status word (LfgcFwdArgs::status) and the host
// entry answers by redoing the launch with the exact-fp32 build, whose range is fp32's (lfgc_capi_forward.hip).
//
// Issue schedule.  v_mfma_f32_32x32x16_f16 occupies the matrix pipe for 32 cycles and the SIMD's vector issue for 8 of
// them; about five independent VALU instructions issue in its shadow (tools/microbench/mfma_shadow.hip: 32.0 -> 33.5
// cycles per MFMA with 0 -> 5 fillers, +4.5 cycles for every further one; with two waves per SIMD the limit is the
// same per MFMA).  Each layer therefore runs as one stream of "gaps" = one MFMA + one slice of other work, written
// out gap by gap with a scheduling fence after each: an A-operand read for a later k-step (ds_read_b128), now and then
// a 1-KiB piece of the next layer's weight stream (lfgc_dma_piece), and 3-4 VALU instructions of the activation + hi/lo
// split of the PREVIOUS output tile -- dependency levels of different accumulator pairs, so that the instructions of
// one gap do not wait for each other (LfgcEpilogue, lfgc_epilogue_gap).  The activation of a layer's last tile rides
// under the first MFMAs of the next layer (whose last two k-steps are the ones that need it).
#pragma once
#include <utility>
#include "lfgc_forward.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void lfgc_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression
template <int N, class F>
__device__ __forceinline__ void lfgc_static_for(F&& f) {
    lfgc_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// hi = rne_f16 of a pair (one v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned lfgc_cvt_pk(float x0, float x1) {
    f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2));
}
// lo = rne_f16(x - f32(hi)) of a pair: ONE mixed-precision FMA per value that reads the f16 half in place and writes
// its f16 result into the destination half (hi * -1 + x is exact in fp32, so the only rounding is the final one)
__device__ __forceinline__ unsigned lfgc_lo_pk(unsigned h, float x0, float x1) {
    unsigned l;        // one statement: between two asm statements hipcc puts an s_nop (an issue slot per pair)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "v"(x0), "v"(x1));
    return l;
}

// x[0..8) fp32 -> hi / lo f16 fragments, 1.5 VALU instructions per value.
__device__ __forceinline__ void lfgc_split8(const float* __restrict__ x, h16x8& hi, h16x8& lo) {
    u32x4 hp, lp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        hp[t] = lfgc_cvt_pk(x[2 * t], x[2 * t + 1]);
        lp[t] = lfgc_lo_pk(hp[t], x[2 * t], x[2 * t + 1]);
    }
    hi = __builtin_bit_cast(h16x8, hp);
    lo = __builtin_bit_cast(h16x8, lp);
}

// Reduced-precision build (LFGC_PRECISION_F16): the hi half only, one packed conversion per pair.
__device__ __forceinline__ void lfgc_cvt8(const float* __restrict__ x, h16x8& hi) {
    u32x4 hp;
#pragma unroll
    for (int t = 0; t < 4; ++t) hp[t] = lfgc_cvt_pk(x[2 * t], x[2 * t + 1]);
    hi = __builtin_bit_cast(h16x8, hp);
}

// Activation + hand-over of one finished 32-row output tile: 8 accumulator pairs x 6 dependency LEVELS.  Pair P =
// accumulator registers 2P, 2P+1 = rows 32m + 8(P>>1) + 4hh + 2(P&1) + {0,1} of the lane's sample; it becomes 32-bit
// element P&3 of output fragment P>>2 of the tile.
//   level 0: t = acc / scale + b/pi (2 fma)          [bias quads stream through LDS reads one quad ahead]
//   level 1: cos(2 pi t) (2 v_cos_f32)
//   level 2: bounded part (1 - cos)/2, scaled (2 fma)
//   level 3: + linear part (2 fma)
//   level 4: f16 hi pair (1 cvt_pk)                   | head: first product (LAST)
//   level 5: f16 lo pair (2 fma_mix)                  | head: second product (LAST)
// Every level depends on the one before it.  An in-order wave that issues them back to back waits out each result
// (one wave per SIMD ran the layers at 62-73 cycles per MFMA gap that way, profiles/r3/stamps_w4_before.log), so the
// levels of one pair go into CONSECUTIVE gaps and each gap carries levels of different pairs -- mutually independent
// instructions -- on the schedule of lfgc_epilogue_gap below.
// The bounded part (1 - cos)/2 = sin(a)^2 is formed first and the linear part added last: the only rounding at the
// magnitude of the result is the final one, like the reference's own 0.5 a + sin(a)^2.
#define LFGC_EP_LEVELS 6
template <bool STASH, bool LAST, bool SPLIT>
struct LfgcEpilogue {
    float inv_scale;
    const float* bias;        // LDS: b/pi of the tile's rows for this lane half (quad q at bias + 8 q)
    const float* wf;          // LDS: final-layer weights of the same rows (LAST)
    float* stash;             // this tile's stash rows for this lane, or nullptr
    f32x4 bq[2], wq[4];
    float t[8][2], v[8][2];   // per pair: pre-activation in turns; cos -> bounded part -> activation (in place)
    unsigned hp[8];

    __device__ __forceinline__ void begin() {                 // issue the first bias quad (call >= 1 gap before level 0)
        bq[0] = *reinterpret_cast<const f32x4*>(bias);
        if (LAST) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const f32x4*>(wf + 8 * q);
        }
    }

    template <int P, int LV>
    __device__ __forceinline__ void level(const f32x16& acc, u32x4 (&Ohi)[2], u32x4 (&Olo)[2], float& ydot, float& tmax) {
        constexpr int Q = P >> 1, I = 2 * (P & 1);
        if constexpr (LV == 0) {
            if constexpr ((P & 1) == 0 && Q + 1 < 4) bq[(Q + 1) & 1] = *reinterpret_cast<const f32x4*>(bias + 8 * (Q + 1));
            t[P][0] = __builtin_fmaf(acc[2 * P], inv_scale, bq[Q & 1][I]);
            t[P][1] = __builtin_fmaf(acc[2 * P + 1], inv_scale, bq[Q & 1][I + 1]);
            if (STASH) {                                                  // the backward kernels read a = pi t
                stash[(2 * P) * 64] = t[P][0] * 3.14159274101257324f;
                stash[(2 * P + 1) * 64] = t[P][1] * 3.14159274101257324f;
            }
            if (LAST) tmax = lfgc_absmax3(tmax, t[P][0], t[P][1]);
        } else if constexpr (LV == 1) {
            v[P][0] = __builtin_amdgcn_cosf(t[P][0]);                    // cos(2 pi t)
            v[P][1] = __builtin_amdgcn_cosf(t[P][1]);
        } else if constexpr (LV == 2) {
            v[P][0] = __builtin_fmaf(v[P][0], -LFGC_ACT_HALF, LFGC_ACT_HALF);
            v[P][1] = __builtin_fmaf(v[P][1], -LFGC_ACT_HALF, LFGC_ACT_HALF);
        } else if constexpr (LV == 3) {
            v[P][0] = __builtin_fmaf(t[P][0], LFGC_ACT_C, v[P][0]);      // scaled, also for the head
            v[P][1] = __builtin_fmaf(t[P][1], LFGC_ACT_C, v[P][1]);
        } else if constexpr (LV == 4) {
            if (LAST) {       // fp32 head on the scaled activations (its weights carry 1 / LFGC_ACT_SCALE)
                ydot = __builtin_fmaf(wq[Q][I], v[P][0], ydot);
            } else {
                hp[P] = lfgc_cvt_pk(v[P][0], v[P][1]);
                Ohi[P >> 2][P & 3] = hp[P];
            }
        } else {
            if (LAST) ydot = __builtin_fmaf(wq[Q][I + 1], v[P][1], ydot);
            else if (SPLIT) Olo[P >> 2][P & 3] = lfgc_lo_pk(hp[P], v[P][0], v[P][1]);
        }
    }

    // Everything at once, level-major (no MFMAs to ride under: the last tile of the last layer, or nets of two k-steps).
    __device__ __forceinline__ void all(const f32x16& acc, u32x4 (&Ohi)[2], u32x4 (&Olo)[2], float& ydot, float& tmax) {
        lfgc_static_for<LFGC_EP_LEVELS>([&](auto lv_c) {
            lfgc_static_for<8>([&](auto p_c) {
                this->template level<decltype(p_c)::value, decltype(lv_c)::value>(acc, Ohi, Olo, ydot, tmax);
            });
        });
    }
};

// The slice of a pending epilogue that rides in gap g of GA.  Gap 0 only issues the first bias read (the accumulator's
// last MFMA has just been issued: reading it at once would wait out the matrix pipe, and the bias its LDS latency); the
// levels run on time steps tau = 0 .. T-1 (T = max(GA - 1, 6)): pair P runs its level lv at tau = s_P + lv with the
// pairs' starts s_P = (T - 6) P / 7 spread over the tile, and step tau belongs to gap 1 + tau (GA - 1) / T -- with
// GA >= 7 one step per gap (the levels of a pair in consecutive gaps, a gap holding the levels of up to three pairs),
// fewer gaps take several steps each.  GA = 1: everything in the one gap.
template <int GA, int g, class EPI>
__device__ __forceinline__ void lfgc_epilogue_gap(EPI& ep, const f32x16& eacc, u32x4 (&Ehi)[2], u32x4 (&Elo)[2],
                                                  float& ydot, float& tmax) {
    if constexpr (g == 0) ep.begin();
    if constexpr (GA == 1) {
        ep.all(eacc, Ehi, Elo, ydot, tmax);
    } else if constexpr (g > 0) {
        constexpr int GE = GA - 1, ge = g - 1;
        constexpr int T = GE > LFGC_EP_LEVELS ? GE : LFGC_EP_LEVELS;
        constexpr int tau_lo = (ge * T + GE - 1) / GE, tau_hi = ((ge + 1) * T + GE - 1) / GE;
        lfgc_static_for<tau_hi - tau_lo>([&](auto d_c) {
            constexpr int tau = tau_lo + decltype(d_c)::value;
            lfgc_static_for<8>([&](auto p_c) {
                constexpr int P = decltype(p_c)::value;
                constexpr int lv = tau - (T - LFGC_EP_LEVELS) * P / 7;
                if constexpr (lv >= 0 && lv < LFGC_EP_LEVELS) ep.template level<P, lv>(eacc, Ehi, Elo, ydot, tmax);
            });
        });
    }
}

// A-operand queue: the fragments of the next LFGC_PF k-steps, read from LDS that many k-steps ahead of their MFMAs.
#ifndef LFGC_PF
#define LFGC_PF 2
#endif
struct LfgcOperands {
    h16x8 hi[LFGC_PF], lo[LFGC_PF];
};

// The gaps of one output tile: G = KS16 * (SPLIT ? 3 : 1) MFMAs accumulating into `acc` (started from 0: the bias is
// added in the epilogue), each followed by the A-operand read of the next k-step (or of `arow_next`'s first) and by its
// slice of the pending epilogue `ep` of accumulator `eacc` (lfgc_epilogue_gap: spread over the first GA gaps; GA = 0: nothing
// pending).  The epilogue writes fragments Ehi / Elo; when these are IN's own last two (a tile carried over from the
// previous layer, GA = gaps of the first KS16 - 2 k-steps) they are copied into IN before k-step KS16 - 2 reads them.
// `w` holds the operands of this tile's first LFGC_PF k-steps on entry and of the next tile's on exit.
// Weight streaming: the NP pieces this wave owes of the NEXT block (lfgc_dma_piece; NP = 0: resident build) are
// requested one at a time in gaps DG0 + g < DSPAN of the layer, evenly spread.
template <int KS16, bool SPLIT, int GA, bool E_IS_IN_TAIL, int NP, int NVEC, int DG0, int DSPAN, int WAVES, class EPI>
__device__ __forceinline__ void lfgc_tile_gaps(const float* __restrict__ arow, const float* __restrict__ arow_next,
                                               u32x4 (&INhi)[KS16], u32x4 (&INlo)[KS16], f32x16& acc,
                                               LfgcOperands& w, EPI& ep, const f32x16& eacc,
                                               u32x4 (&Ehi)[2], u32x4 (&Elo)[2], float& ydot, float& tmax,
                                               const LfgcDmaPlan& dma) {
    constexpr int MPK = SPLIT ? 3 : 1;
    static_assert(!E_IS_IN_TAIL || GA <= (KS16 - 2) * MPK, "a carried tile must be done before the k-steps that read it");
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    lfgc_static_for<KS16>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (E_IS_IN_TAIL && ks == KS16 - 2) {
            INhi[KS16 - 2] = Ehi[0]; INhi[KS16 - 1] = Ehi[1];
            INlo[KS16 - 2] = Elo[0]; INlo[KS16 - 1] = Elo[1];
        }
        const h16x8 whi = w.hi[0], wlo = w.lo[0];          // this k-step's operands (read LFGC_PF k-steps ago)
        h16x8 nhi = w.hi[LFGC_PF - 1], nlo = w.lo[LFGC_PF - 1];
        lfgc_static_for<MPK>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            constexpr int g = ks * MPK + u;
            const h16x8 bh = __builtin_bit_cast(h16x8, INhi[ks]);
            if (SPLIT) {
                const h16x8 bl = __builtin_bit_cast(h16x8, INlo[ks]);
                if (u == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh, acc, 0, 0, 0);       // small terms first
                if (u == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bl, acc, 0, 0, 0);
                if (u == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh, acc, 0, 0, 0);
            }
            if constexpr (NP > 0 && !(LFGC_ABLATE & 64)) {   // this gap's share of the weight stream, in the MFMA's shadow (diagnostics: 64 = none)
                constexpr int gg = DG0 + g;
                constexpr int pi_lo = gg < DSPAN ? (gg * NP + DSPAN - 1) / DSPAN : NP;
                constexpr int pi_hi = gg + 1 < DSPAN ? ((gg + 1) * NP + DSPAN - 1) / DSPAN : NP;
                lfgc_static_for<(gg < DSPAN ? pi_hi - pi_lo : 0)>([&](auto p_c) { lfgc_dma_piece_ct<WAVES, pi_lo + decltype(p_c)::value, NVEC>(dma); });
            }
            // the operands of k-step ks + LFGC_PF (running on into the next tile's rows), one read per gap
            const float* nsrc = (ks + LFGC_PF < KS16) ? arow + 16 * (ks + LFGC_PF)
                                                      : (arow_next ? arow_next + 16 * (ks + LFGC_PF - KS16) : nullptr);
#if !(LFGC_ABLATE & 32)             // diagnostics: 32 = no operand reads after the first k-steps'
            if (nsrc) {
                if (u == 0) nhi = *reinterpret_cast<const h16x8*>(nsrc);
                if (SPLIT && u == 1) nlo = *reinterpret_cast<const h16x8*>(nsrc + 4);
            }
#endif
            if constexpr (GA > 0 && g < GA && !(LFGC_ABLATE & 16))        // diagnostics: 16 = no epilogue under the MFMAs
                lfgc_epilogue_gap<GA, g>(ep, eacc, Ehi, Elo, ydot, tmax);
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int d = 0; d + 1 < LFGC_PF; ++d) { w.hi[d] = w.hi[d + 1]; w.lo[d] = w.lo[d + 1]; }
        w.hi[LFGC_PF - 1] = nhi; w.lo[LFGC_PF - 1] = nlo;
    });
}

// What a layer leaves for the next one to finish: the accumulator of its last output tile and that tile's constants.
struct LfgcCarry {
    f32x16 acc;
    float inv_scale;
    const float* bias;
    float* stash;
};

// One hidden layer on a 32-sample tile.  IN = KS16 input fragments (the last two still owed by `carry` when
// HAS_CARRY: they are produced under this layer's first MFMAs), OUT = this layer's 2*MT output fragments (all but the
// last two; those are `out_carry`'s to produce), or with LAST the head's partial dot product in `ydot` (complete).
template <int KS16, int MT, int S, bool STASH, bool LAST, bool SPLIT, bool HAS_CARRY, int NP, int NVEC, int WAVES>
__device__ __forceinline__ void lfgc_layer_fwd16(const float* __restrict__ s_blk, u32x4 (&INhi)[KS16], u32x4 (&INlo)[KS16],
                                                 const LfgcCarry& carry, float inv_scale, const float* __restrict__ s_bias,
                                                 u32x4 (&OUThi)[2 * MT], u32x4 (&OUTlo)[2 * MT], LfgcCarry& out_carry,
                                                 const float* __restrict__ s_final, float& ydot, float& tmax,
                                                 float* __restrict__ stash, int j, int hh, int lane, const LfgcDmaPlan& dma) {
    constexpr int MPK = SPLIT ? 3 : 1;
    constexpr int G = KS16 * MPK;
    // the next block's pieces go out over the first half of the layer's gaps (1/4 ... 1 measured alike, profiles/r3/ab_dma_policy_span.log)
#ifndef LFGC_DMA_SPAN4
#define LFGC_DMA_SPAN4 2
#endif
    constexpr int DSPAN = (MT * G * LFGC_DMA_SPAN4 + 3) / 4;
    const float* s_row = s_blk + j * S + 8 * hh;              // lane half hh: bytes [32 hh, 32 hh + 32) of each 64-B k-step
    const float* bias_l = s_bias + 4 * hh;
    const float* wf_l = s_final + 4 * hh;
    float* stash_l = nullptr;
    if (STASH) { stash_l = stash + lane; asm volatile("" : "+v"(stash_l)); }
    LfgcOperands w;
#pragma unroll
    for (int d = 0; d < LFGC_PF; ++d) {                        // (KS16 >= LFGC_PF for every compiled shape)
        w.hi[d] = *reinterpret_cast<const h16x8*>(s_row + 16 * d);
        w.lo[d] = w.hi[d];
        if (SPLIT) w.lo[d] = *reinterpret_cast<const h16x8*>(s_row + 16 * d + 4);
    }

    f32x16 accs[2];
    {   // tile 0: shadows the previous layer's last tile, which produces this layer's last two input fragments
        LfgcEpilogue<STASH, false, SPLIT> ep;
        u32x4 ehi[2], elo[2];
        constexpr int GA0 = HAS_CARRY ? (KS16 - 2) * MPK : 0;
        if (HAS_CARRY) {
            ep.inv_scale = carry.inv_scale; ep.bias = carry.bias; ep.wf = nullptr; ep.stash = carry.stash;
            if constexpr (GA0 == 0) {      // two k-steps in all (H <= 32): both read the owed fragments, nothing to hide under
                ep.begin();
                ep.all(carry.acc, ehi, elo, ydot, tmax);
                INhi[KS16 - 2] = ehi[0]; INhi[KS16 - 1] = ehi[1]; INlo[KS16 - 2] = elo[0]; INlo[KS16 - 1] = elo[1];
            }
        }
        lfgc_tile_gaps<KS16, SPLIT, GA0, (HAS_CARRY && GA0 > 0), NP, NVEC, 0, DSPAN, WAVES>(s_row, MT > 1 ? s_row + 32 * S : nullptr, INhi, INlo,
                                                    accs[0], w, ep, carry.acc, ehi, elo, ydot, tmax, dma);
    }
    // tiles 1 .. MT-1: each shadows the tile before it
    lfgc_static_for<MT - 1>([&](auto m_c) {
        constexpr int m = 1 + decltype(m_c)::value;
        LfgcEpilogue<STASH, LAST, SPLIT> ep;
        ep.inv_scale = inv_scale; ep.bias = bias_l + 32 * (m - 1); ep.wf = wf_l + 32 * (m - 1);
        ep.stash = STASH ? stash_l + (m - 1) * (16 * 64) : nullptr;
        u32x4 ehi[2], elo[2];
        lfgc_tile_gaps<KS16, SPLIT, G, false, NP, NVEC, m * G, DSPAN, WAVES>(s_row + 32 * m * S, m + 1 < MT ? s_row + 32 * (m + 1) * S : nullptr,
                                              INhi, INlo, accs[m & 1], w, ep, accs[(m - 1) & 1], ehi, elo, ydot, tmax, dma);
        if (!LAST) {
            OUThi[2 * (m - 1)] = ehi[0]; OUThi[2 * (m - 1) + 1] = ehi[1];
            OUTlo[2 * (m - 1)] = elo[0]; OUTlo[2 * (m - 1) + 1] = elo[1];
        }
    });
    // the last tile: finished here for the head, otherwise owed to the next layer
    if (LAST) {
        LfgcEpilogue<STASH, true, SPLIT> ep;
        ep.inv_scale = inv_scale; ep.bias = bias_l + 32 * (MT - 1); ep.wf = wf_l + 32 * (MT - 1);
        ep.stash = STASH ? stash_l + (MT - 1) * (16 * 64) : nullptr;
        u32x4 ehi[2], elo[2];
        ep.begin();
        ep.all(accs[(MT - 1) & 1], ehi, elo, ydot, tmax);
    } else {
        out_carry.acc = accs[(MT - 1) & 1];
        out_carry.inv_scale = inv_scale;
        out_carry.bias = bias_l + 32 * (MT - 1);
        out_carry.stash = STASH ? stash_l + (MT - 1) * (16 * 64) : nullptr;
    }
}

// Diagnostics (-DLFGC_STAMPS): cycle stamps between the phases of a tile, summed per wave.  In the shipped build no
// stamp executes.
#ifdef LFGC_STAMPS
#define LFGC_STAMP(k) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); st_acc[k] += now__ - st_last; st_last = now__; } while (0)
#else
#define LFGC_STAMP(k) do { } while (0)
#endif

// ZRUN (lattice mode only, never with STASH): z-run tiles and the column sampler (LfgcColumnSampler, lfgc_forward.h).
template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH, bool SPLIT, bool ZRUN>
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
    // 1-KiB DMA pieces per wave for a hidden layer's block (streamed under the layer before it) and for layer 0's (under
    // the last layer of the batch before)
    constexpr int NP1 = STREAM ? ((BLK1 / 4 + 63) / 64 + WAVES - 1) / WAVES : 0;
    constexpr int NP0 = STREAM ? ((BLK0 / 4 + 63) / 64 + WAVES - 1) / WAVES : 0;
    // offsets inside the packed blob (lfgc_common.h)
    constexpr int F_BLK0 = HP * (K0P + 4) + HP, F_BLK1 = HP * (HP + 4) + HP;
    constexpr int K0R = (K0P + 31) / 32 * 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_final = smem;               // Wf (HP) | bf (4)
    float* s_scale = smem + HP + 4;      // scale[8] | 1/scale[8]
    float* s_bias = s_scale + 16;        // b/pi of every hidden layer: LFGC_MAX_LAYERS x HP
    float* s_w = s_bias + LFGC_MAX_LAYERS * HP;   // resident: every layer block; streamed: ring of 2 x BLKMAX
    float* s_coord = s_w + (STREAM ? 2 * BLKMAX : (BLK0 + (a.L - 1) * BLK1));
    float* s_col = s_coord + ((a.res0 + a.res1 + a.res2 + 3) & ~3) + (threadIdx.x >> 6) * (a.nzc * (CH + 4));   // ZRUN: this wave's column (rows of LfgcColumnSampler::CS floats)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // the same number, known to be wave-uniform
    const int j = lane & 31;
    const int hh = lane >> 5;
    const int L = a.L;
    const int off_final = F_BLK0 + (L - 1) * F_BLK1;
    const int off_h = off_final + HP + 4 + K0R * (HP + 4) + (L - 1) * HP * (HP + 4);
    const float* hblk = a.packed + off_h + 32 + LFGC_MAX_LAYERS * HP + HP;
    {
        // head: weights divided by LFGC_ACT_SCALE (the last hidden layer's activations arrive scaled), bias as it is
        for (int i = tid; i < HP; i += NT) s_final[i] = a.packed[off_h + 32 + LFGC_MAX_LAYERS * HP + i];
        if (tid < 4) s_final[HP + tid] = a.packed[off_final + HP + tid];
        if (tid < 16) s_scale[tid] = a.packed[off_h + 16 + tid];       // the forward images' own scales
        for (int i = tid; i < L * HP; i += NT) s_bias[i] = a.packed[off_h + 32 + i];
        if (!STREAM) {
            const f32x4* srcw = reinterpret_cast<const f32x4*>(hblk);
            for (int i = tid; i < (BLK0 + (L - 1) * BLK1) / 4; i += NT) reinterpret_cast<f32x4*>(s_w)[i] = srcw[i];
        } else {
            lfgc_dma_to_lds(hblk, s_w, BLK0, wave, lane, WAVES);
#if LFGC_ABLATE & 8             // diagnostics: both ring slots hold layer 1's block for good (finite data, wrong results)
            lfgc_dma_to_lds(hblk + BLK0, s_w, BLK1, wave, lane, WAVES);
            lfgc_dma_to_lds(hblk + BLK0, s_w + BLKMAX, BLK1, wave, lane, WAVES);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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
#ifdef LFGC_PRIO_YOUNG      // diagnostics: static priority for the second-dispatched half of the workgroup
    if (WAVES == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#if (LFGC_ABLATE & 8) && defined(LFGC_ANTIPHASE)
    // diagnostics: with the barriers gone, start the second wave of every SIMD LFGC_ANTIPHASE x 8k cycles late
    if (wave >= WAVES / 2) {
#pragma unroll 1
        for (int i = 0; i < LFGC_ANTIPHASE; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
#ifdef LFGC_STAMPS
    unsigned long long st_acc[16] = {0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    const long long N = a.n;
    // ZRUN: tile t = (row t / tiles_per_row of the slab, z run t % tiles_per_row); the walk over a workgroup's tiles is kept
    // in wave-uniform (x, y, run) form, advanced by the grid stride with carries -- no division per batch
    int zr_x = 0, zr_y = 0, zr_t = 0, zr_dx = 0, zr_dy = 0, zr_dt = 0;
    if (ZRUN) {
        const unsigned tpr = (unsigned)a.tiles_per_row, r1 = (unsigned)a.res1;
        const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + wave));
        const unsigned row0 = t0 / tpr, stride = gridDim.x * WAVES, drow = stride / tpr;
        zr_t = (int)(t0 - row0 * tpr); zr_x = (int)(row0 / r1); zr_y = (int)(row0 - (row0 / r1) * r1);
        zr_dt = (int)(stride - drow * tpr); zr_dx = (int)(drow / r1); zr_dy = (int)(drow - (drow / r1) * r1);
    }
    for (long long batch = blockIdx.x; batch < a.nbatches; batch += gridDim.x) {
        const long long tile_idx = batch * WAVES + wave;
        long long n = tile_idx * LFGC_TILE_SAMPLES + j;
        bool valid = n < N;
        const long long nc = valid ? n : (N - 1);
        LFGC_STAMP(0);            // loop overhead / previous tail

        LfgcSampler<CH, NF> sampler;
        LfgcColumnSampler<CH, NF> csampler;
#if !(LFGC_ABLATE & 128)           // diagnostics: 128 = no input phase at all (inputs = a cheap function of the lane)
        if constexpr (ZRUN) {
            const bool tile_ok = tile_idx < a.ntiles;             // (a tile past the end repeats the slab's first row, unstored)
            const int vx = tile_ok ? zr_x : 0, vy = tile_ok ? zr_y : 0, vz = (tile_ok ? zr_t : 0) * LFGC_TILE_SAMPLES + j;
            valid = tile_ok && vz < a.res2;
            n = ((long long)vx * a.res1 + vy) * a.res2 + vz;
            csampler.stage_a(a, a.x_begin + vx, vy, min(vz, a.res2 - 1), s_coord, s_col, lane);
            zr_t += zr_dt;
            const int c1 = zr_t >= a.tiles_per_row ? 1 : 0;
            zr_t -= c1 * a.tiles_per_row;
            zr_y += zr_dy + c1;
            const int c2 = zr_y >= a.res1 ? 1 : 0;
            zr_y -= c2 * a.res1;
            zr_x += zr_dx + c2;
        } else {
            sampler.issue(a, nc, N, s_coord, hh);       // 8 corner rows requested; used after the layer-0 barrier
        }
#endif

        LfgcDmaPlan dma = {hblk, s_w, BLK0 / 4, wave_s, (unsigned)lane * 16u, 0ull, 0u};
        // opaque per batch: otherwise the address arithmetic of every piece of every block is hoisted out of this loop
        // into ~60 SGPRs that do not exist (spilled to VGPR lanes, v_readlane in the gaps)
        asm volatile("" : "+s"(dma.wave));
        auto acquire = [&](int l) -> const float* {
            if (!STREAM) return s_w + (l == 0 ? 0 : BLK0 + (l - 1) * BLK1);
            LFGC_STAMP(2 + 2 * (l < 6 ? l : 6));           // the layer before this acquire (or the input phase for l = 0 -> slot 1 below)
#if LFGC_ABLATE & 8                                        // diagnostics: no weight streaming, no barriers (wrong results)
            return s_w + (l & 1) * BLKMAX;
#endif
            // my pieces of this layer's block have landed (they are older than the sampler's loads, which stay in
            // flight across the layer-0 barrier: vmcnt counts in issue order)
#ifndef LFGC_LATE_BARRIER0
            // (ZRUN: the column sampler's corner quads, 4 per pass, are the youngest loads)
            if (l == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ZRUN ? 4 * LfgcColumnSampler<CH, NF>::NPASS : 8 * (CH / 8)) : "memory");
            else
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LFGC_STAMP(1);                                  // wait for my DMA pieces
            __syncthreads();
            LFGC_STAMP(3 + 2 * (l < 5 ? l : 5));            // barrier before layer l
            const float* blk = s_w + (step & 1) * BLKMAX;
            // the next block (the layer after this one, or layer 0 of the next batch) goes into the other ring slot -- every
            // wave is past the barrier, so nobody reads it any more -- piece by piece under this layer's MFMAs
            // (after the last batch the slot is filled once more for nobody: an unconditional stream keeps the gaps
            // free of branches)
            const int ln = (l + 1 == L) ? 0 : l + 1;
            dma.src = hblk + (ln == 0 ? 0 : BLK0 + (long long)(ln - 1) * BLK1);
            dma.dst = s_w + ((step + 1) & 1) * BLKMAX;
            dma.nvec = (ln == 0 ? BLK0 : BLK1) / 4;
            lfgc_dma_plan_block(dma);
            ++step;
            return blk;
        };
        float* stash_tile = STASH ? a.stash + tile_idx * (long long)(64 * (KS0 + L * 16 * MT)) + 64 * KS0 : nullptr;
        auto stash_of = [&](int l) -> float* { return STASH ? stash_tile + (long long)l * (64 * 16 * MT) : nullptr; };

        float ydot = 0.0f, tmax = 0.0f;
        u32x4 Ahi[2 * MT], Alo[2 * MT], Bhi[2 * MT], Blo[2 * MT];
        LfgcCarry ca, cb;
        {   // layer 0
#ifndef LFGC_LATE_BARRIER0
            const float* blk = acquire(0);
#endif
            float X[8 * KS16_0];
            {
                float B0[KS0];
#if !(LFGC_ABLATE & 128)
                if constexpr (ZRUN) csampler.stage_b(s_col, hh, B0);
                else sampler.finish(hh, B0);
#else
#pragma unroll
                for (int s = 0; s < KS0; ++s) B0[s] = 0.01f * (float)(lane + s);
#endif
#pragma unroll
                for (int s = 0; s < KS0; ++s) X[s] = B0[s];
#pragma unroll
                for (int s = KS0; s < 8 * KS16_0; ++s) X[s] = 0.0f;
            }
            if (STASH) {
                float* px = a.stash + tile_idx * (long long)(64 * (KS0 + L * 16 * MT)) + lane;
                asm volatile("" : "+v"(px));
#pragma unroll
                for (int s = 0; s < KS0; ++s) px[s * 64] = X[s];
            }
            u32x4 X0hi[KS16_0], X0lo[KS16_0];
#pragma unroll
            for (int s = 0; s < KS16_0; ++s) {
                h16x8 fh, fl;
                if (SPLIT) lfgc_split8(X + 8 * s, fh, fl);
                else { lfgc_cvt8(X + 8 * s, fh); fl = fh; }
                X0hi[s] = __builtin_bit_cast(u32x4, fh); X0lo[s] = __builtin_bit_cast(u32x4, fl);
            }
#ifdef LFGC_LATE_BARRIER0
            const float* blk = acquire(0);
#endif
            if (L == 1)
                lfgc_layer_fwd16<KS16_0, MT, S0, STASH, true, SPLIT, false, NP0, BLK0 / 4, WAVES>(blk, X0hi, X0lo, ca, s_scale[8], s_bias, Ahi, Alo, ca,
                                                                            s_final, ydot, tmax, stash_of(0), j, hh, lane, dma);
            else
                lfgc_layer_fwd16<KS16_0, MT, S0, STASH, false, SPLIT, false, NP1, BLK1 / 4, WAVES>(blk, X0hi, X0lo, ca, s_scale[8], s_bias, Ahi, Alo, ca,
                                                                             s_final, ydot, tmax, stash_of(0), j, hh, lane, dma);
        }
        // hidden layers 1 .. L-2 in ping-pong pairs, then the last one with the head folded in
        {
            int l = 1;
            for (; l + 2 < L; l += 2) {
                const float* blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, false, SPLIT, true, NP1, BLK1 / 4, WAVES>(blk, Ahi, Alo, ca, s_scale[8 + l], s_bias + l * HP, Bhi, Blo, cb,
                                                                            s_final, ydot, tmax, stash_of(l), j, hh, lane, dma);
                blk = acquire(l + 1);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, false, SPLIT, true, NP1, BLK1 / 4, WAVES>(blk, Bhi, Blo, cb, s_scale[9 + l], s_bias + (l + 1) * HP, Ahi, Alo, ca,
                                                                            s_final, ydot, tmax, stash_of(l + 1), j, hh, lane, dma);
            }
            if (l + 1 < L) {       // one more non-final layer: A -> B, final consumes B
                const float* blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, false, SPLIT, true, NP1, BLK1 / 4, WAVES>(blk, Ahi, Alo, ca, s_scale[8 + l], s_bias + l * HP, Bhi, Blo, cb,
                                                                            s_final, ydot, tmax, stash_of(l), j, hh, lane, dma);
                ++l;
                blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, true, SPLIT, true, NP0, BLK0 / 4, WAVES>(blk, Bhi, Blo, cb, s_scale[8 + l], s_bias + l * HP, Ahi, Alo, ca,
                                                                           s_final, ydot, tmax, stash_of(l), j, hh, lane, dma);
            } else if (l < L) {    // final layer consumes A
                const float* blk = acquire(l);
                lfgc_layer_fwd16<KS16_1, MT, S1, STASH, true, SPLIT, true, NP0, BLK0 / 4, WAVES>(blk, Ahi, Alo, ca, s_scale[8 + l], s_bias + l * HP, Bhi, Blo, cb,
                                                                           s_final, ydot, tmax, stash_of(l), j, hh, lane, dma);
            }
        }

        LFGC_STAMP(14);               // last layer
        float y = ydot + __shfl_xor(ydot, 32);
        y += s_final[HP];
        // range screen (header comment): an f16 overflow anywhere upstream has made y NaN; the last layer's
        // pre-activations are checked explicitly.  A sample the fast arithmetic cannot do is reported, never returned
        // as a finite wrong number.
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (!(tmax <= LFGC_TURNS_MAX)) y = __builtin_nanf("");
        if (a.status && valid && !(__builtin_fabsf(y) < __builtin_inff()))
            __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.clamp) y = fminf(fmaxf(y, -1.0f), 1.0f);
        if (valid && hh == 0) a.out[n] = y;
    }
    // the stream's last block (fetched for nobody) must have landed before the workgroup gives its LDS back
    if (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LFGC_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* dst = a.stamps + ((long long)blockIdx.x * WAVES + wave) * 20;
        for (int k = 0; k < 16; ++k) dst[k] = st_acc[k];
        dst[16] = __builtin_amdgcn_s_memtime() - st_t0;
        dst[17] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
}

template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH, bool SPLIT, bool ZRUN>
static int lfgc_launch_fwd16_cfg(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    auto kern = lfgc_fwd16_kernel<CH, MT, NF, WAVES, STREAM, STASH, SPLIT, ZRUN>;
    // the >64 KB dynamic-LDS attribute is per device: raised once per (instantiation, device); one-time, so launches
    // stay graph-capturable
    static int lds_limit_set[LFGC_MAX_DEVICES] = {0};
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

template <int CH, int MT, int NF, int WAVES, bool STREAM, bool STASH>
static int lfgc_launch_fwd16_one(const LfgcFwdArgs& a, int lds_bytes, int grid, hipStream_t stream) {
    if constexpr (!STASH) {
        if (a.zrun)
            return a.single ? lfgc_launch_fwd16_cfg<CH, MT, NF, WAVES, STREAM, false, false, true>(a, lds_bytes, grid, stream)
                            : lfgc_launch_fwd16_cfg<CH, MT, NF, WAVES, STREAM, false, true, true>(a, lds_bytes, grid, stream);
    }
    return a.single ? lfgc_launch_fwd16_cfg<CH, MT, NF, WAVES, STREAM, STASH, false, false>(a, lds_bytes, grid, stream)
                    : lfgc_launch_fwd16_cfg<CH, MT, NF, WAVES, STREAM, STASH, true, false>(a, lds_bytes, grid, stream);
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
