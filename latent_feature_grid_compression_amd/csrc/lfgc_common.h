// lfgc_common.h -- shared device helpers and the packed-parameter plan (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lfgc.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LFGC_WAVE 64
// f16-split forward (lfgc_forward16.h): activations handed to the next layer are multiplied by LFGC_ACT_SCALE so that
// their f16 hi half overflows before the pre-activation leaves the domain of v_cos_f32 (|a / pi| <= 256):
// s ((pi/2) 256 - 1) = 65582 > 65520 (rounds to inf), s ((pi/2) 250 + 1) = 64368 < 65504 (largest finite f16).
// The value is chosen so that both constants of the scaled activation  s h = (s pi/2) t + s/2 - (s/2) cos(2 pi t)
// are fp32 numbers: s pi/2 = LFGC_ACT_C exactly and s/2 = c/pi = LFGC_ACT_HALF to 1.8e-12 -- no systematic error in
// the linear term (a rounded pi/2 would add 2.8e-8 |a|/2 to every activation with the same sign).
#define LFGC_ACT_C 256.8191833496094f              // 0x1.00d1b6p+8
#define LFGC_ACT_HALF 81.74808502197266f           // 0x1.46fe0ap+6 = LFGC_ACT_C / pi
#define LFGC_ACT_SCALE 163.49617004365646          // double: LFGC_ACT_C / (pi / 2); divisor of the next layer's weight image
#define LFGC_TURNS_MAX 255.0f            // last layer (no f16 conversion): explicit screen
#define LFGC_TILE_SAMPLES 32          // samples per wave tile (N dimension of v_mfma_f32_32x32x2_f32)

// ------------------------------------------------------------------------------------------------
// Packed-parameter plan.  Everything is derived from the descriptor; host and device agree through
// these constexpr-able formulas.
//   CH   grid channels padded to 8            MT   hidden width in 32-row MFMA tiles
//   E    scalar inputs 3 + 6*NF               EP   E padded to 8
//   K0P  padded layer-0 fan-in = CH + EP      KS0  layer-0 k-steps = K0P/2 (one MFMA = 2 k values)
//   HP   padded hidden = 32*MT                KS1  hidden k-steps = HP/2
// Row strides carry +4 floats so that the 16 lanes of a ds_read_b128 group (distinct rows, same
// column) fall on 16 distinct 16-byte bank slots (stride = 4 * odd).
// Blob (floats):  [layer0: W (HP x S0) | bias HP] [layer l=1..L-1: W (HP x S1) | bias HP]
//                 [final: Wf HP | bf 4]
//                 [transposed copies for backward: layer0^T (K0R x ST, K0R = K0P rounded to 32), layer l^T (HP x ST)]
// ------------------------------------------------------------------------------------------------
struct LfgcPlan {
    int C, CH, H, HP, MT, L, NF, E, EP, K0P, K0R, KS0, KS1, S0, S1, ST;
    int blk0, blk1;          // floats per forward layer block (weights + bias)
    int off_final;           // float offset of [Wf | bf]
    int fwd_floats;          // floats of the forward part
    int tblk0, tblk1;        // floats per transposed block
    int off_t;               // float offset of the transposed part
    int total_floats;
    int stash_tile_floats;   // floats saved per 32-sample tile: 64 * (KS0 + L*16*MT)
    // f16-split forward section (lfgc_forward16.h): per-layer power-of-two scales, then layer blocks whose rows
    // hold, per 16-column k-step, [lane half 0: 8 hi halfs | 8 lo halfs][lane half 1: 8 hi | 8 lo] (64 B), i.e.
    // 4 bytes per weight like the fp32 blocks, same +16 B row padding, followed by the scaled fp32 bias.
    int K0P16, SH0, SH1, blkh0, blkh1, off_h, off_hbias, off_hwf, off_hblk;
    int off_ht;              // f16-split TRANSPOSED images for the backward data chain (same sizes as tblk0 / tblk1)
};

__host__ __device__ inline int lfgc_roundup(int v, int m) { return (v + m - 1) / m * m; }

__host__ __device__ inline LfgcPlan lfgc_make_plan(int C, int H, int L, int NF) {
    LfgcPlan p;
    p.C = C; p.H = H; p.L = L; p.NF = NF;
    p.CH = lfgc_roundup(C, 8);
    p.HP = lfgc_roundup(H, 32);
    if (p.HP == 96) p.HP = 128;          // compiled tile counts: MT in {1, 2, 4}
    p.MT = p.HP / 32;
    p.E = 3 + 6 * NF;
    p.EP = lfgc_roundup(p.E, 8);
    p.K0P = p.CH + p.EP;
    p.KS0 = p.K0P / 2;
    p.KS1 = p.HP / 2;
    p.S0 = p.K0P + 4;
    p.S1 = p.HP + 4;
    p.ST = p.HP + 4;
    p.blk0 = p.HP * p.S0 + p.HP;
    p.blk1 = p.HP * p.S1 + p.HP;
    p.off_final = p.blk0 + (L - 1) * p.blk1;
    p.fwd_floats = p.off_final + p.HP + 4;
    p.K0R = lfgc_roundup(p.K0P, 32);     // layer-0 transposed image: rows padded to whole 32-row MFMA tiles
    p.tblk0 = p.K0R * p.ST;
    p.tblk1 = p.HP * p.ST;
    p.off_t = p.fwd_floats;
    p.K0P16 = lfgc_roundup(p.K0P, 16);
    p.SH0 = p.K0P16 + 4;
    p.SH1 = p.HP + 4;
    p.blkh0 = p.HP * p.SH0 + p.HP;
    p.blkh1 = p.HP * p.SH1 + p.HP;
    // 32 floats: scale[8] | 1/scale[8] of the transposed images (true W), then scale[8] | 1/scale[8] of the forward
    // images (W / pi [/ LFGC_ACT_SCALE], lfgc_forward16.h); then the hidden-layer biases divided by pi, un-scaled,
    // LFGC_MAX_LAYERS x HP (the f16-split forward adds them in its epilogue and keeps them LDS-resident); then the head's
    // weights divided by LFGC_ACT_SCALE (the last hidden layer hands its activations over scaled like every other)
    p.off_h = p.off_t + p.tblk0 + (L - 1) * p.tblk1;
    p.off_hbias = p.off_h + 32;
    p.off_hwf = p.off_hbias + LFGC_MAX_LAYERS * p.HP;        // head weights divided by LFGC_ACT_SCALE: HP floats
    p.off_hblk = p.off_hwf + p.HP;
    p.off_ht = p.off_hblk + p.blkh0 + (L - 1) * p.blkh1;
    p.total_floats = p.off_ht + p.tblk0 + (L - 1) * p.tblk1;
    p.stash_tile_floats = 64 * (p.KS0 + L * 16 * p.MT);
    return p;
}

// Original nn.Linear column of layer 0 that packed column `cl` (0..K0P) holds, or -1 for padding.
// Packed order: k-step s = cl/8*4 + cl%4, lane half hh = (cl/4)&1.  Steps s < CH/2 carry grid
// channel hh*CH/2 + s; later steps carry scalar input hh*EP/2 + (s - CH/2) of
// [p0 p1 p2 | sin f0 (3) cos f0 (3) | sin f1 ... ]  (model/Feature_Grid_Model.py:69 column order:
// [input(3), embedding(6*NF), features(C)]).
__host__ __device__ inline int lfgc_layer0_src_col(const LfgcPlan& p, int cl) {
    const int s = (cl >> 3) * 4 + (cl & 3);
    const int hh = (cl >> 2) & 1;
    if (s < p.CH / 2) {
        const int ch = hh * (p.CH / 2) + s;
        return ch < p.C ? p.E + ch : -1;
    }
    const int e = hh * (p.EP / 2) + (s - p.CH / 2);
    return e < p.E ? e : -1;
}

// f16-split blocks: original nn.Linear column held by (k-step b, lane half hp, element jj) of layer l, or -1.
// Hidden layers: the k order of a 32x32x16 MFMA step follows the accumulator rows a lane holds in registers
// 8s..8s+7 of tile m (b = 2m + s): column 32m + 16s + (jj&3) + 8(jj>>2) + 4hp.  Layer 0: lane half hp feeds its
// own input list X[8b + jj] = [CH/2 grid channels | EP/2 scalars | zero pad]  (lfgc_sample_inputs).
__host__ __device__ inline int lfgc_h16_src_col(const LfgcPlan& p, int l, int b, int hp, int jj) {
    if (l > 0) {
        const int c = 16 * b + (jj & 3) + 8 * (jj >> 2) + 4 * hp;
        return c < p.H ? c : -1;
    }
    const int idx = 8 * b + jj;
    if (idx < p.CH / 2) {
        const int ch = hp * (p.CH / 2) + idx;
        return ch < p.C ? p.E + ch : -1;
    }
    if (idx < p.CH / 2 + p.EP / 2) {
        const int e = hp * (p.EP / 2) + (idx - p.CH / 2);
        return e < p.E ? e : -1;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// sin / cos for the embedding and SnakeAlt: Cody-Waite reduction by pi (3 fp32 constants) to
// r in [-pi/2, pi/2], then sin r = r + r^3 P(r^2), cos r = 1 + r^2 Q(r^2); max abs error 1.4e-7 for
// |x| <= 2^15 (measured against fp64; tests/test_hip_kernels.py).  Outside that range (a diverged model) the
// exact-reduction libm path is used.  The hardware v_sin_f32 (~1e-6 abs) is not accurate enough for
// the 1e-5 end-to-end parity budget.
// ------------------------------------------------------------------------------------------------
#define LFGC_TRIG_FAST_MAX 32768.0f

// r = x - k*pi with pi = hi + mid (+ 3.4e-15 dropped: k <= 10431 inside the fast range, error < 4e-11)
__device__ __forceinline__ float lfgc_reduce_pi_nosign(float x) {
    const float k = __builtin_rintf(x * 0.31830987334251404f);
    float r = __builtin_fmaf(-k, 3.14159274101257324f, x);          // fp32(pi)
    return __builtin_fmaf(-k, -8.74227765734758577e-08f, r);        // fp32(pi - hi)
}

__device__ __forceinline__ float lfgc_reduce_pi_fast(float x, float& sign_bits) {
    const float k = __builtin_rintf(x * 0.31830987334251404f);
    float r = __builtin_fmaf(-k, 3.14159274101257324f, x);
    r = __builtin_fmaf(-k, -8.74227765734758577e-08f, r);
    const int ki = (int)k;
    sign_bits = __int_as_float(ki << 31);                           // (-1)^k as a sign bit
    return r;
}

// Rare path (|x| > 2^15, inf, nan): the same reduction carried in fp64 with a two-term pi; accurate to
// fp32 rounding while k = rint(x/pi) is exact in fp64 (|x| < ~1e15), finite garbage in [-1,1] beyond,
// NaN for inf/nan inputs like libm.
__device__ __forceinline__ float lfgc_reduce_pi_wide(float x, float& sign_bits) {
    const double xd = (double)x;
    const double k = __builtin_rint(xd * 0.31830988618379067154);
    double r = __builtin_fma(-k, 3.141592653589793116, xd);
    r = __builtin_fma(-k, 1.2246467991473532072e-16, r);
    const double half = k * 0.5;
    sign_bits = (half != __builtin_floor(half)) ? -0.0f : 0.0f;
    return (float)r;
}

// true when x must take the wide path (|x| > 2^15, inf or nan)
__device__ __forceinline__ bool lfgc_trig_out_of_range(float x) { return !(__builtin_fabsf(x) <= LFGC_TRIG_FAST_MAX); }

__device__ __forceinline__ float lfgc_sin_poly(float r) {
    const float u = r * r;
    float p = 2.6340962904214393e-06f;
    p = __builtin_fmaf(p, u, -0.00019822562171611935f);
    p = __builtin_fmaf(p, u, 0.008333241567015648f);
    p = __builtin_fmaf(p, u, -0.1666666567325592f);
    return __builtin_fmaf(r * u, p, r);
}

__device__ __forceinline__ float lfgc_cos_poly(float r) {
    const float u = r * r;
    float q = -2.628940194426832e-07f;
    q = __builtin_fmaf(q, u, 2.47742427745834e-05f);
    q = __builtin_fmaf(q, u, -0.0013888647081330419f);
    q = __builtin_fmaf(q, u, 0.0416666604578495f);
    q = __builtin_fmaf(q, u, -0.5f);
    return __builtin_fmaf(u, q, 1.0f);
}

// WIDE = false: branch-free fast path (caller guarantees or separately checks the range);
// WIDE = true : fp64 reduction, valid for every input.  Hot loops run the fast form on a whole tile and
// redo the tile with the wide form under ONE wave-uniform branch if any lane was out of range, so the
// MFMA/VALU stream is not cut into basic blocks per activation.
template <bool WIDE>
__device__ __forceinline__ float lfgc_sinf_t(float x) {
    float sb;
    const float r = WIDE ? lfgc_reduce_pi_wide(x, sb) : lfgc_reduce_pi_fast(x, sb);
    return __int_as_float(__float_as_int(lfgc_sin_poly(r)) ^ __float_as_int(sb));
}

template <bool WIDE>
__device__ __forceinline__ float lfgc_cosf_t(float x) {
    float sb;
    const float r = WIDE ? lfgc_reduce_pi_wide(x, sb) : lfgc_reduce_pi_fast(x, sb);
    return __int_as_float(__float_as_int(lfgc_cos_poly(r)) ^ __float_as_int(sb));
}

template <bool WIDE>
__device__ __forceinline__ void lfgc_sincosf_t(float x, float& s, float& c) {
    float sb;
    const float r = WIDE ? lfgc_reduce_pi_wide(x, sb) : lfgc_reduce_pi_fast(x, sb);
    s = __int_as_float(__float_as_int(lfgc_sin_poly(r)) ^ __float_as_int(sb));
    c = __int_as_float(__float_as_int(lfgc_cos_poly(r)) ^ __float_as_int(sb));
}

// SnakeAlt(a) = 0.5 a + sin(a)^2            (model/Feature_Grid_Model.py:12-13)
// sin(a)^2 does not depend on the (-1)^k sign of the reduced sine, so the fast form skips it.
template <bool WIDE>
__device__ __forceinline__ float lfgc_snake_t(float a) {
    float s;
    if (WIDE) {
        s = lfgc_sinf_t<true>(a);
    } else {
        s = lfgc_sin_poly(lfgc_reduce_pi_nosign(a));
    }
    return __builtin_fmaf(s, s, 0.5f * a);
}

// Tile-level range screen for the fast path: running max of |x| (NaN/inf need no screening: they propagate
// to NaN through the fast path exactly as libm does).
__device__ __forceinline__ float lfgc_absmax3(float m, float a, float b) {
    float r;   // one instruction; fmaxf() would add canonicalising v_max per operand
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

// Asynchronous global -> LDS copy of `nfloats` (multiple of 4) floats laid out identically on both sides
// (LDS-DMA: no VGPR staging; each wave-instruction moves 64 lanes x 16 B to wave-uniform base + lane*16).
// Completion: s_waitcnt vmcnt(0) in the issuing wave, then a workgroup barrier before any wave reads.
__device__ __forceinline__ void lfgc_dma_to_lds(const float* __restrict__ gsrc, float* lds_dst, int nfloats,
                                                int wave, int lane, int nwaves) {
    const int nvec = nfloats >> 2;
    for (int base = wave * 64; base < nvec; base += nwaves * 64) {
        const int i = base + lane;
        if (i < nvec) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(gsrc + 4 * i),
                (__attribute__((address_space(3))) void*)(lds_dst + 4 * base), 16, 0, 0);
        }
    }
}

// The same copy cut into its 1-KiB pieces (one wave-instruction each) so that a caller can spread them over its own
// instruction stream: piece `pi` of this wave = piece wave + pi * nwaves of the block.  A wave that issues its 8-9 pieces
// of a 68-KB layer block back to back -- as every wave of the workgroup does at the same moment after the layer's
// barrier -- queues behind the CU's one texture-address path: measured ~170 cycles per piece, 1.4 k cycles per wave and
// layer (profiles/r3/stamps_w8_before.log: layer time - 96 x 64).  One piece every few MFMA gaps issues in the matrix
// pipe's shadow instead (lfgc_dma_piece below).  (Requesting the pieces in a different order per CU -- so that the CUs of an XCD do not ask the
// L2 for the same line at once -- was measured too: 2.7 % SLOWER, profiles/r3/ab_epilogue_dma.log.)
// Branch-free on purpose (a predicated piece would cut the caller's MFMA stream into basic blocks): a piece index past
// the end of the block, and the partial last piece, are clamped onto the block's last 64 vectors -- the same bytes to
// the same LDS addresses again, harmless -- and a plan always names a real block (when no batch follows, the caller
// lets the unused slot be filled once more).
struct LfgcDmaPlan {
    const float* src;     // global block
    float* dst;           // LDS block (same layout)
    int nvec;             // 16-byte vectors in the block (>= 64)
    int wave;             // wave-uniform (readfirstlane)
    unsigned lane16;      // lane * 16
    unsigned long long src_w;   // set by lfgc_dma_plan_block: address of this wave's first piece ...
    unsigned dst_w;             // ... and its LDS byte address
};
// Per block, once: the wave's own base addresses, from which lfgc_dma_piece_ct forms a piece's with constants.
__device__ __forceinline__ void lfgc_dma_plan_block(LfgcDmaPlan& d) {
    const unsigned long long ga = (unsigned long long)(size_t)(d.src + 4 * (d.wave << 6));
    d.src_w = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga) |
              ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32)) << 32);
    d.dst_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(d.dst + 4 * (d.wave << 6)));
}
template <int NWAVES>
__device__ __forceinline__ void lfgc_dma_piece(const LfgcDmaPlan& d, int pi) {
    int base = (d.wave + pi * NWAVES) << 6;              // wave-uniform
    base = base < d.nvec - 64 ? base : d.nvec - 64;
    const unsigned long long ga = (unsigned long long)(size_t)(d.src + 4 * base);
    // (wave-uniform by construction; readfirstlane makes it so for the register allocator where it cannot prove it)
    const unsigned long long sp = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga) |
                                  ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32)) << 32);
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(d.dst + 4 * base));   // low half of the flat address = LDS offset
    // Written out: the builtin is given a 64-bit per-lane address, and hipcc then keeps one VGPR pair per piece alive
    // across the whole batch loop (spilled, and reloaded behind an s_waitcnt vmcnt(0) in the middle of the MFMA stream).
    // This form takes the block address from SGPRs and one 32-bit lane offset that never changes.  M0 is not named as a
    // clobber (hipcc reserves it); nothing else in the kernels that use this sets M0 except the builtin form of the same
    // instruction, which writes M0 itself right before each use.
#ifndef LFGC_DMA_POLICY
#define LFGC_DMA_POLICY ""
#endif
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" LFGC_DMA_POLICY :: "s"(lds_addr), "v"(d.lane16), "s"(sp) : "memory");
}

// Piece PI (a compile-time index) of a block of NVEC vectors (compile-time too).  Where every wave's piece PI lies wholly
// inside the block -- all but the last one or two -- its addresses are the wave's block bases (lfgc_dma_plan_block) plus
// constants: 3 scalar instructions and the load, against 10 for the clamped form.  A lone instruction stream gets one
// issue slot per 4 cycles for scalar instructions as for vector ones.
template <int NWAVES, int PI, int NVEC>
__device__ __forceinline__ void lfgc_dma_piece_ct(const LfgcDmaPlan& d) {
    if constexpr ((NWAVES - 1 + PI * NWAVES) * 64 + 64 <= NVEC) {
        const unsigned long long sp = d.src_w + (unsigned long long)PI * NWAVES * 1024ull;
        const unsigned lds_addr = d.dst_w + (unsigned)PI * NWAVES * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" LFGC_DMA_POLICY :: "s"(lds_addr), "v"(d.lane16), "s"(sp) : "memory");
    } else {
        lfgc_dma_piece<NWAVES>(d, PI);
    }
}

// d SnakeAlt / da = 0.5 + 2 sin a cos a
template <bool WIDE>
__device__ __forceinline__ float lfgc_snake_grad_t(float a) {
    float s, c;
    lfgc_sincosf_t<WIDE>(a, s, c);
    return __builtin_fmaf(2.0f * s, c, 0.5f);
}

// fp32 frequency f_k = fp32(2^k) * 2 * pi evaluated the way torch does for a float tensor
// (model/Feature_Embedding.py:28-29): (2^k * 2) is exact, times fp32-rounded... torch multiplies the
// fp32 tensor by the python double 2*pi?  No: `freq_bands * 2. * np.pi` = (t * 2.) * 3.14159..., each a
// tensor-scalar multiply carried out in fp32 with the scalar cast to fp32 -> fp32(2^(k+1)) * fp32(pi).
__device__ __forceinline__ float lfgc_freq(int k) {
    return (float)(2 << k) * 3.14159274101257324f;     // exact power of two times fp32(pi): one rounding, exact here
}

// Per-device host-side caches (CU count, one-time kernel attributes) are keyed by the current HIP device: a process may
// drive several devices (the caller selects the device of its tensors before calling in; ops.py does).
#define LFGC_MAX_DEVICES 16
static inline int lfgc_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LFGC_MAX_DEVICES) dev = 0;
    return dev;
}
static inline int lfgc_num_cus() {
    static int cus[LFGC_MAX_DEVICES] = {0};
    const int dev = lfgc_current_device();
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

#define LFGC_HIP_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
