// lfgc_wavelet_cl.hip -- the LAST wavelet level with the dense grid in the sampler's channel-last layout (gfx950).
//   lfgc_idwt_level_cl_f32      wavelet_transform/Torch_Wavelet_Transform.py:91-104 (+ crop :69-73) writing (t0,t1,t2,Cs)
//   lfgc_idwt_level_cl_bwd_f32  its adjoint reading the gradient of that (t0,t1,t2,Cs) grid
// so that decode_volume() needs no layout conversion pass (lfgc_grid_layout_f32) on either direction.  Separable
// filter banks only (`taps`); the dense-stencil path keeps the channel-first kernels + the conversion.
//
// Both kernels have one shape.  A workgroup owns 32 consecutive cells of the flattened (y,x) plane, a group of CW = 32,
// 16 or 8 channels (CW / 2 waves), and walks a chunk of z steps.  The coefficient side is channel-first (contiguous along the cells), the
// grid side channel-last (contiguous along the channels), so the arithmetic runs with the lanes along whichever side is
// being READ -- straight from global memory, neighbouring lanes on neighbouring addresses, the 4x reuse between
// neighbouring cells served by L1/L2 -- and the 8 results per (cell, channel) go through an LDS tile that is read back
// with the lanes along the other side: every global store instruction writes whole 64/128-byte runs.  The tile is double
// buffered: one barrier per z step; the stores of step s and the loads of step s+1 are in flight under the arithmetic
// (a thread holds exactly one step's 32 input values; the next step's loads are issued as soon as they are contracted).
// Along z the stencil slides: a coefficient plane (synthesis) / a pair of source planes (adjoint) is read and contracted
// in-plane ONCE; what it contributes to the following step is carried in 8 registers per (cell, channel).
// Work items (plane tile, channel group, z chunk) are dealt to the XCDs so that the two channel groups of a tile (they
// write the two halves of the same 128-byte lines) and neighbouring tiles (they share coefficient rows / source voxels)
// run on the same XCD and meet in its L2.
#include "lfgc_common.h"
#include <cstdlib>

namespace {

constexpr int kCells = 32;           // plane cells per workgroup

template <int CW> struct ClShape {   // CW: channels per workgroup (8, 16; synthesis also 32)
    static constexpr int NW = CW / 2;                    // waves per workgroup
    static constexpr int CPW = 64 / CW;                  // voxels / cells per wave instruction with the lanes along channels
    static constexpr int VOX = CW + 1;                   // synthesis tile [8 parities][32 cells][CW + 1]
    static constexpr int CHS = 8 * kCells + 1;           // adjoint tile [CW channels][8 bands][32 cells] + 1
};

// Loads go through buffer descriptors: a 32-bit lane offset on a wave-uniform (descriptor, scalar offset) pair costs one
// VGPR per distinct lane offset instead of a 64-bit address per load, and a lane offset >= num_records reads as 0.0 --
// taps outside the level (the conv_transpose3d / F.pad zeros) are encoded as kOutside in the lane offset: no select per
// value, no address clamp.  The scalar offset is not part of the range check, so it may be any in-range plane offset.
// A thread holds one step's 32 input values; the next step's are requested as soon as these are contracted and fly
// under the rest of the step (the carry arithmetic, the tile, the barrier, the stores).  Requesting them a whole step
// ahead (two register sets, each path through a step with its own wait counts) was built and measured: same time at
// 114 instead of 96 VGPRs (synthesis), slower where it cost a workgroup per CU (adjoint: 146 VGPRs) -- the kernels
// are not latency-bound.
constexpr unsigned kOutside = 0x40000000u;            // arrays on these paths are < 2^30 bytes (host check)
typedef __amdgpu_buffer_rsrc_t cl_srd;
#ifndef LFGC_CL_ABLATE
#define LFGC_CL_ABLATE 0             // diagnostics (tools/ab_wavelet_cl.py): 1 no global stores, 2 no global loads
#endif

__device__ __forceinline__ cl_srd cl_make_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void cl_load(float& dst, cl_srd r, unsigned lane_off, unsigned uniform_off) {
    if (LFGC_CL_ABLATE & 2) { dst = __builtin_bit_cast(float, (lane_off ^ uniform_off) & 0x3fffffu); return; }
    dst = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)uniform_off, 0));
}

// NT: non-temporal (streaming) store.  The synthesis writes whole 64/128-byte runs of a grid nobody reads before the
// kernel ends; keeping them out of the L2 leaves it to the coefficient lines that neighbouring workgroups share
// (measured, cfg-5 last level: 205 -> 180 us; only with whole lines, i.e. 32-channel workgroups: 64- and 32-byte
// pieces get slower).  Only for grids that could not stay in the 32 MB of L2 anyway: the cfg-3 grid (32 MiB) is sampled
// right after the decode and the train step is 4 us faster with it cached (0.405 vs 0.409 ms).  The adjoint's stores
// are 128-byte pieces of unaligned runs that complete each other's lines in the L2: streaming them costs (192 ->
// 236 us), and so does streaming any of the loads.
template <bool NT>
__device__ __forceinline__ void cl_store(float v, cl_srd r, unsigned lane_off, unsigned uniform_off) {
    if ((LFGC_CL_ABLATE & 1) && v != 1.2345e-30f) return;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)lane_off, (int)uniform_off, NT ? 2 : 0);   // lane_off >= num_records: dropped
}

// Workgroup b works on item (b % 8) * ceil(total / 8) + b / 8 (workgroups are dealt round-robin to the 8 XCDs);
// item = (z chunk * ptiles + plane tile) * ngroups + channel group.
__device__ __forceinline__ bool cl_work_item(int ptiles, int ngroups, int nchunks, int* pt, int* cg, int* zc) {
    const int total = ptiles * ngroups * nchunks;
    const int per = (total + 7) >> 3;
    const int item = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= total) return false;
    const int tile = item / ngroups;
    *cg = item - tile * ngroups;
    *zc = tile / ptiles;
    *pt = tile - *zc * ptiles;
    return true;
}

struct IdwtClArgs {
    const float* lll;   // (C, d0,d1,d2)
    const float* hf;    // (C, 7, d0,d1,d2)
    float* out;         // (t0,t1,t2, cs)
    int C, cs, d0, d1, d2, t0, t1, t2, o0, o1, o2;   // o = crop offset floor((2d+2-t)/2)
    int zchunk, ptiles, ngroups, nchunks;
    float taps[8];      // [low | high][tap]
};

// Synthesis: out_full[o] = sum_{s,t} in[s][i] F_s[t], o = 2 i + t per axis; cell j = (jz,jy,jx) in [0,d] per axis
// produces the 2x2x2 outputs o = 2 j + p from the coefficient cells i = j - e (e in {0,1}) with taps t = p + 2 e.
// Plane iz of the coefficients is contracted over x and y once (Y[sz][py][px]); its e_z = 0 part completes cell slice
// jz = iz (added to the carry of plane iz - 1), its e_z = 1 part is the carry for slice iz + 1.
// Arithmetic role: 32 cells x 2 channels per wave: channel = c0 + 2 wave + lane / 32.
template <int CW, int CELLS, bool NT>
__global__ __launch_bounds__(CW * CELLS) void idwt_cl_kernel(const IdwtClArgs a) {
    constexpr int CPL = 64 / CELLS;                   // channels per wave in the arithmetic role
    constexpr int CPW = ClShape<CW>::CPW, VOX = ClShape<CW>::VOX;
    constexpr int TILE = 8 * CELLS * VOX;
    extern __shared__ __attribute__((aligned(16))) float s_tile[];      // [2][TILE]
    int pt, cg, zc;
    if (!cl_work_item(a.ptiles, a.ngroups, a.nchunks, &pt, &cg, &zc)) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = a.d0 + 1, n1 = a.d1 + 1, n2 = a.d2 + 1;
    const int plane_cells = n1 * n2;
    const int f0 = pt * CELLS;
    const int c0 = cg * CW;
    const int jz_begin = zc * a.zchunk, jz_end = min(jz_begin + a.zchunk, n0);
    const int dplane = a.d1 * a.d2;
    const int dvol = dplane * a.d0;                   // C * 7 * dvol * 4 < 2^30 (host check)

    // arithmetic role: byte offsets of the 4 neighbour cells (jy - ey, jx - ex) inside a coefficient plane, plus the
    // lane's channel parity; everything else of the address is wave-uniform
    const int cell = lane & (CELLS - 1), chalf = lane / CELLS;
    const int cw = CPL * w + chalf;                      // channel within the group
    unsigned offl[4];                                  // low band; the detail bands' offset is offl + hshift (7 bands per channel)
    unsigned hshift;
    {
        const int f = f0 + cell;
        const int fc = min(f, plane_cells - 1);
        const int jy = fc / n2, jx = fc - jy * n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cy = jy - (q >> 1), cx = jx - (q & 1);
            const bool ok = f < plane_cells && c0 + cw < a.C && cy >= 0 && cy < a.d1 && cx >= 0 && cx < a.d2;
            offl[q] = ok ? 4u * (unsigned)(chalf * dvol + cy * a.d2 + cx) : kOutside;
        }
        hshift = 4u * (unsigned)(chalf * 6 * dvol);        // kOutside + hshift stays >= num_records: 6 * dvol * 4 < 2^30
    }
    const cl_srd rl = cl_make_srd(a.lll, (unsigned)(a.C * dvol * 4));
    const cl_srd rh = cl_make_srd(a.hf, (unsigned)(a.C * 7 * dvol * 4));
    const bool wave_live = c0 + CPL * w < a.C;           // wave-uniform: this wave's channel pair exists

    // store role: lanes = CW channels x CPW voxels; this wave's cells are [CPW w, CPW w + CPW), instruction p = parity:
    // one cell per thread
    const int fch = lane & (CW - 1), fcell = w * CPW + lane / CW;
    unsigned vo[4];                                    // byte offset of output (py,px) of the cell in a z slice; kOutside when cropped away
    {
        const int f = f0 + fcell;
        const int fc = min(f, plane_cells - 1);
        const int jy = fc / n2, jx = fc - jy * n2;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int oy = 2 * jy + (p >> 1) - a.o1, ox = 2 * jx + (p & 1) - a.o2;
            const bool ok = f < plane_cells && c0 + fch < a.cs && oy >= 0 && oy < a.t1 && ox >= 0 && ox < a.t2;
            vo[p] = ok ? 4u * (unsigned)((oy * a.t2 + ox) * a.cs + c0 + fch) : kOutside;
        }
    }
    const bool pad_channel = c0 + fch >= a.C;
    const int slice = a.t1 * a.t2 * a.cs;
    const cl_srd rout = cl_make_srd(a.out, (unsigned)(a.t0 * slice * 4));

    float R[32];                                       // [neighbour q][band]
    float carry[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) carry[p] = 0.0f;

    auto issue = [&](int iz) {                         // coefficient plane iz: 4 neighbour cells x 8 bands of this lane's channel
        const unsigned sl = 4u * (unsigned)((c0 + CPL * w) * dvol + iz * dplane);
#pragma unroll
        for (int q = 0; q < 4; ++q) cl_load(R[q * 8], rl, offl[q], sl);
#pragma unroll
        for (int sb = 1; sb < 8; ++sb) {
            const unsigned sh = 4u * (unsigned)(((c0 + CPL * w) * 7 + sb - 1) * dvol + iz * dplane);
#pragma unroll
            for (int q = 0; q < 4; ++q) cl_load(R[q * 8 + sb], rh, offl[q] + hshift, sh);
        }
    };

    int iz = jz_begin - 1;
    if (wave_live && iz >= 0) issue(iz);
    int buf = 0;
#pragma unroll 1
    for (; iz < jz_end; ++iz) {
        const bool emit = iz >= jz_begin;
        const bool plane_ok = iz >= 0 && iz < a.d0;
        const bool next_ok = iz + 1 < jz_end && iz + 1 < a.d0;
        float* tile = s_tile + buf * TILE;
        if (wave_live) {
            float outv[8];
            if (plane_ok) {
                float X[2][2][2][2];                              // [ey][sz][sy][px]
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int ey = q >> 2, sz = (q >> 1) & 1, sy = q & 1;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        float t = 0.0f;
#pragma unroll
                        for (int ex = 0; ex < 2; ++ex)
#pragma unroll
                            for (int sx = 0; sx < 2; ++sx)
                                t = __builtin_fmaf(R[(ey * 2 + ex) * 8 + sz * 4 + sy * 2 + sx], a.taps[sx * 4 + px + 2 * ex], t);
                        X[ey][sz][sy][px] = t;
                    }
                }
                if (next_ok) issue(iz + 1);                       // R is free: the next plane flies under the rest of the step
                float Y[2][2][2];                                 // [sz][py][px]
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int sz = q >> 2, py = (q >> 1) & 1, px = q & 1;
                    float t = 0.0f;
#pragma unroll
                    for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                        for (int sy = 0; sy < 2; ++sy)
                            t = __builtin_fmaf(X[ey][sz][sy][px], a.taps[sy * 4 + py + 2 * ey], t);
                    Y[sz][py][px] = t;
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int pz = p >> 2, py = (p >> 1) & 1, px = p & 1;
                    float t = carry[p];                           // plane iz - 1 (e_z = 1)
                    float n = 0.0f;
#pragma unroll
                    for (int sz = 0; sz < 2; ++sz) {
                        t = __builtin_fmaf(Y[sz][py][px], a.taps[sz * 4 + pz], t);
                        n = __builtin_fmaf(Y[sz][py][px], a.taps[sz * 4 + pz + 2], n);
                    }
                    outv[p] = t;
                    carry[p] = n;
                }
            } else {
                if (next_ok) issue(iz + 1);
#pragma unroll
                for (int p = 0; p < 8; ++p) { outv[p] = carry[p]; carry[p] = 0.0f; }
            }
            if (emit) {
#pragma unroll
                for (int p = 0; p < 8; ++p) tile[(p * CELLS + cell) * VOX + cw] = outv[p];
            }
        }
        if (emit) {
            __syncthreads();
            // one barrier per step is enough: the other buffer is written only after the NEXT barrier, which every wave
            // reaches after it has finished reading this one's predecessor
#pragma unroll
            for (int pz = 0; pz < 2; ++pz) {
                const int oz = 2 * iz + pz - a.o0;
                if (oz >= 0 && oz < a.t0) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float v = tile[((pz * 4 + p) * CELLS + fcell) * VOX + fch];
                        cl_store<NT>(pad_channel ? 0.0f : v, rout, vo[p], 4u * (unsigned)(oz * slice));
                    }
                }
            }
            buf ^= 1;
        }
    }
}

struct AnalysisClArgs {
    const float* src;      // (n0,n1,n2, cs)
    float* band0;          // band 0 of channel c at band0 + c * dvol
    float* bandh;          // band s >= 1 of channel c at bandh + (c * 7 + s - 1) * dvol
    int C, cs, n0, n1, n2, lo0, lo1, lo2, d0, d1, d2;
    int zchunk, ptiles, ngroups, nchunks;
    float taps[8];
};

// Adjoint: band_s[c][i] = sum_t src[2 i + t - lo][c] F_s[t].  Step iz reads the source planes 2 iz - lo0 + {2, 3},
// contracts each over x and y (P[sy][sx]) and combines them with the two planes carried from step iz - 1.
// Arithmetic role: lanes = CW channels x CPW cells; cell = wave * CPW + lane / CW.
template <int CW, int NG>          // NG: cell groups of 32 per workgroup (the channel-first runs it writes are 128 NG bytes)
__global__ __launch_bounds__(32 * CW) void analysis_cl_kernel(const AnalysisClArgs a) {
    constexpr int CPW = ClShape<CW>::CPW, CELLS = kCells * NG, CHS = 8 * CELLS + 1;
    constexpr int TILE = CW * CHS;
    extern __shared__ __attribute__((aligned(16))) float s_tile[];      // [2][TILE]: [channel][band][cell] + 1
    int pt, cg, zc;
    if (!cl_work_item(a.ptiles, a.ngroups, a.nchunks, &pt, &cg, &zc)) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int plane_cells = a.d1 * a.d2;
    const int f0 = pt * CELLS;
    const int c0 = cg * CW;
    const int iz_begin = zc * a.zchunk, iz_end = min(iz_begin + a.zchunk, a.d0);
    const int dvol = plane_cells * a.d0;              // C * 7 * dvol * 4 < 2^30 (host check)
    const int nplane = a.n1 * a.n2 * a.cs;            // n0 * nplane * 4 < 2^30 (host check)

    // arithmetic role: lanes = CW channels x CPW cells; cell of group g = 32 g + wave * CPW + lane / CW
    const int ch = lane & (CW - 1), cslot = w * CPW + lane / CW;
    unsigned ro[NG][4], co[NG][4]; // byte offsets of row ty / column tx (+ channel) inside a source plane; kOutside when outside
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int f = f0 + g * kCells + cslot;
        const bool live = f < plane_cells && c0 + ch < a.C;
        const int fc = min(f, plane_cells - 1);
        const int iy = fc / a.d2, ix = fc - iy * a.d2;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int uy = 2 * iy + t - a.lo1, ux = 2 * ix + t - a.lo2;
            ro[g][t] = (live && uy >= 0 && uy < a.n1) ? 4u * (unsigned)(uy * a.n2 * a.cs) : kOutside;
            co[g][t] = (ux >= 0 && ux < a.n2) ? 4u * (unsigned)(ux * a.cs + c0 + ch) : kOutside;
        }
    }
    const cl_srd rs = cl_make_srd(a.src, (unsigned)(a.n0 * nplane * 4));

    // store role: NG == 1: lane = (cell, channel parity), one instruction per band; NG >= 2: lane = cell, one instruction per
    // band, channel and 64 cells
    unsigned so0, soh;
    if (NG == 1) {
        const int sf = f0 + (lane & 31), half = lane >> 5;
        const bool ok = sf < plane_cells && c0 + 2 * w + half < a.C;
        so0 = ok ? 4u * (unsigned)(half * dvol + sf) : kOutside;
        soh = ok ? 4u * (unsigned)(half * 7 * dvol + sf) : kOutside;
    } else {
        so0 = soh = 4u * (unsigned)(f0 + lane);
    }
    const cl_srd rb0 = cl_make_srd(a.band0, (unsigned)(a.C * dvol * 4));
    const cl_srd rbh = cl_make_srd(a.bandh, (unsigned)(a.C * 7 * dvol * 4));

    float R[32];                   // [plane k][ty*4+tx] of the cell group in flight
    float carry[NG][2][4];         // [group][plane tz = 0, 1 of the next step][sy*2+sx]
#pragma unroll
    for (int i = 0; i < 8 * NG; ++i) carry[i >> 3][(i >> 2) & 1][i & 3] = 0.0f;

    auto plane_in = [&](int iz, int k) { const int uz = 2 * iz - a.lo0 + 2 + k; return uz >= 0 && uz < a.n0; };
    auto issue = [&](int iz, int g) {                  // source planes 2 iz - lo0 + {2,3} (outside the level: not read, P = 0)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (plane_in(iz, k)) {
                const unsigned sp = 4u * (unsigned)((2 * iz - a.lo0 + 2 + k) * nplane);
#pragma unroll
                for (int t = 0; t < 16; ++t) cl_load(R[k * 16 + t], rs, ro[g][t >> 2] + co[g][t & 3], sp);
            }
        }
    };

    int iz = iz_begin - 1;
    issue(iz, 0);
    int buf = 0;
#pragma unroll 1
    for (; iz < iz_end; ++iz) {
        const bool emit = iz >= iz_begin;
        float* tile = s_tile + buf * TILE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float P[2][4];                                        // new planes tz = 2, 3: [sy*2+sx]
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (plane_in(iz, k)) {
                    float X[4][2];
#pragma unroll
                    for (int ty = 0; ty < 4; ++ty) {
                        float x0 = 0.0f, x1 = 0.0f;
#pragma unroll
                        for (int tx = 0; tx < 4; ++tx) {
                            x0 = __builtin_fmaf(R[k * 16 + ty * 4 + tx], a.taps[tx], x0);
                            x1 = __builtin_fmaf(R[k * 16 + ty * 4 + tx], a.taps[4 + tx], x1);
                        }
                        X[ty][0] = x0; X[ty][1] = x1;
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        float t = 0.0f;
#pragma unroll
                        for (int ty = 0; ty < 4; ++ty) t = __builtin_fmaf(X[ty][s4 & 1], a.taps[(s4 >> 1) * 4 + ty], t);
                        P[k][s4] = t;
                    }
                } else {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) P[k][s4] = 0.0f;
                }
            }
            // R is free: the next group's / step's planes fly under the rest of this one
            if (g + 1 < NG) issue(iz, g + 1);
            else if (iz + 1 < iz_end) issue(iz + 1, 0);
            if (emit) {
#pragma unroll
                for (int sb = 0; sb < 8; ++sb) {
                    const int s4 = sb & 3, sz = sb >> 2;
                    float t = carry[g][0][s4] * a.taps[sz * 4 + 0];
                    t = __builtin_fmaf(carry[g][1][s4], a.taps[sz * 4 + 1], t);
                    t = __builtin_fmaf(P[0][s4], a.taps[sz * 4 + 2], t);
                    t = __builtin_fmaf(P[1][s4], a.taps[sz * 4 + 3], t);
                    tile[ch * CHS + sb * CELLS + g * kCells + cslot] = t;
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { carry[g][0][s4] = P[0][s4]; carry[g][1][s4] = P[1][s4]; }
        }
        if (emit) {
            __syncthreads();
            // store role: this wave's channels are {2 w, 2 w + 1}; whole 128 NG-byte runs of one (channel, band) per half wave / wave
            const unsigned zoff = (unsigned)(iz * plane_cells);
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) {
                if (NG == 1) {                                    // both channels in one instruction (lane / 32)
                    const float v = tile[(2 * w + (lane >> 5)) * CHS + sb * CELLS + (lane & 31)];
                    if (sb == 0) cl_store<false>(v, rb0, so0, 4u * ((unsigned)((c0 + 2 * w) * dvol) + zoff));
                    else cl_store<false>(v, rbh, soh, 4u * ((unsigned)(((c0 + 2 * w) * 7 + sb - 1) * dvol) + zoff));
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int cb = c0 + 2 * w + h;
                        if (cb >= a.C) continue;
#pragma unroll
                        for (int part = 0; part < NG / 2; ++part) {                   // 64 cells = 256 contiguous bytes each
                            const float v = tile[(2 * w + h) * CHS + sb * CELLS + part * 64 + lane];
                            const unsigned lo = (f0 + part * 64 + lane < plane_cells) ? so0 + 256u * part : kOutside;
                            if (sb == 0) cl_store<false>(v, rb0, lo, 4u * ((unsigned)(cb * dvol) + zoff));
                            else cl_store<false>(v, rbh, lo, 4u * ((unsigned)((cb * 7 + sb - 1) * dvol) + zoff));
                        }
                    }
                }
            }
            buf ^= 1;
        }
    }
}

// z chunk length: few enough workgroups per slot that no round is mostly idle, long enough that the warm-up plane of a
// chunk (read and contracted, nothing emitted) stays a small share.
int pick_zchunk(long long columns, int nz, int slots_per_cu) {
    if (const char* e = getenv("LFGC_CL_NCHUNKS")) { const int nc = atoi(e); if (nc >= 1 && nc <= nz) return (nz + nc - 1) / nc; }   // diagnostics
    const int slots = slots_per_cu * lfgc_num_cus();
    long long best_cost = -1;
    int best = nz;
    for (int nchunks = 1; nchunks <= nz; ++nchunks) {
        const int zc = (nz + nchunks - 1) / nchunks;
        const long long groups = columns * ((nz + zc - 1) / zc);
        const long long rounds = (groups + slots - 1) / slots;
        const long long cost = rounds * (zc + 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = zc; }
    }
    return best;
}

template <typename K, typename A>
int launch_cl(K kern, int threads, const A& a, int lds_bytes, hipStream_t stream) {
    const long long total = (long long)a.ptiles * a.ngroups * a.nchunks;
    const long long blocks = (total + 7) / 8 * 8;      // cl_work_item: 8 XCD queues of ceil(total / 8)
    if (blocks > 0x7fffffffLL || lds_bytes > 80 * 1024) return LFGC_E_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds_bytes, stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

int check_cl(const void* p0, const void* p1, const void* p2, const void* p3, const float* taps, int C, int cs,
             int d0, int d1, int d2, int t0, int t1, int t2) {
    if (!p0 || !p1 || !p2 || !p3) return LFGC_E_NULL;
    if (!taps) return LFGC_E_UNSUPPORTED;              // dense stencil: channel-first kernels + lfgc_grid_layout_f32
    if (C < 1 || d0 < 1 || d1 < 1 || d2 < 1 || t0 < 1 || t1 < 1 || t2 < 1) return LFGC_E_SHAPE;
    if (t0 > 2 * d0 + 2 || t1 > 2 * d1 + 2 || t2 > 2 * d2 + 2) return LFGC_E_SHAPE;
    if (cs != lfgc_roundup(C, 8)) return LFGC_E_SHAPE;
    // buffer descriptors with kOutside as the out-of-range marker: every array below 2^30 bytes
    if ((long long)t0 * t1 * t2 * cs * 4 >= (1LL << 30) || (long long)d0 * d1 * d2 * 7 * C * 4 >= (1LL << 30)) return LFGC_E_UNSUPPORTED;
    return LFGC_OK;
}

}  // namespace

extern "C" int lfgc_idwt_level_cl_f32(const float* lll, const float* hf, const float* taps, float* out_cl,
                                      int C, int channel_stride, int d0, int d1, int d2, int t0, int t1, int t2,
                                      lfgc_stream_t stream) {
    const int rc = check_cl(lll, hf, out_cl, out_cl, taps, C, channel_stride, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    IdwtClArgs a = {};
    a.lll = lll; a.hf = hf; a.out = out_cl;
    a.C = C; a.cs = channel_stride; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
    a.o0 = (2 * d0 + 2 - t0) / 2; a.o1 = (2 * d1 + 2 - t1) / 2; a.o2 = (2 * d2 + 2 - t2) / 2;
    for (int i = 0; i < 8; ++i) a.taps[i] = taps[i];
    // 32 channels on a large plane: one workgroup of 16 waves writes whole 128-byte lines (d = 65: 197 vs 203 us); on a
    // small one two 8-wave groups balance better (d = 33: 26.5 vs 28.7 us).  24 channels: three groups of 8.
    const long long ptiles = ((long long)(d1 + 1) * (d2 + 1) + kCells - 1) / kCells;
    int cw = channel_stride == 32 ? (ptiles >= 96 ? 32 : 16) : channel_stride == 16 ? 16 : 8;
    if (const char* e = getenv("LFGC_CL_CW")) { const int v = atoi(e); if ((v == 8 || v == 16 || v == 32) && channel_stride % v == 0) cw = v; }   // diagnostics
    a.ngroups = channel_stride / cw;
    if (ptiles * a.ngroups > 0x0fffffffLL) return LFGC_E_UNSUPPORTED;
    a.ptiles = (int)ptiles;
    const int lds = 2 * 8 * kCells * (cw + 1) * 4;
    a.zchunk = pick_zchunk(ptiles * a.ngroups, d0 + 1, cw == 32 ? 1 : cw == 16 ? 2 : 4);    // 96 VGPRs: 4 waves per SIMD
    a.nchunks = (d0 + 1 + a.zchunk - 1) / a.zchunk;
    hipStream_t st = (hipStream_t)stream;
    bool nt = cw == 32 && (long long)t0 * t1 * t2 * channel_stride * 4 > (48LL << 20); // see cl_store
    if (const char* e = getenv("LFGC_CL_NT")) nt = e[0] == '1';                         // diagnostics
    if (cw == 32) {
        static bool raised[LFGC_MAX_DEVICES] = {false};     // 67.6 KB of LDS: above the 64 KB default limit
        const int dev = lfgc_current_device();
        if (!raised[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(idwt_cl_kernel<32, kCells, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(idwt_cl_kernel<32, kCells, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            raised[dev] = true;
        }
        return nt ? launch_cl(idwt_cl_kernel<32, kCells, true>, 1024, a, lds, st) : launch_cl(idwt_cl_kernel<32, kCells, false>, 1024, a, lds, st);
    }
    if (cw == 16) return nt ? launch_cl(idwt_cl_kernel<16, kCells, true>, 512, a, lds, st) : launch_cl(idwt_cl_kernel<16, kCells, false>, 512, a, lds, st);
    return nt ? launch_cl(idwt_cl_kernel<8, kCells, true>, 256, a, lds, st) : launch_cl(idwt_cl_kernel<8, kCells, false>, 256, a, lds, st);
}

extern "C" int lfgc_idwt_level_cl_bwd_f32(const float* d_out_cl, const float* taps, float* d_lll, float* d_hf,
                                          int C, int channel_stride, int d0, int d1, int d2, int t0, int t1, int t2,
                                          lfgc_stream_t stream) {
    const int rc = check_cl(d_out_cl, d_lll, d_hf, d_hf, taps, C, channel_stride, d0, d1, d2, t0, t1, t2);
    if (rc != LFGC_OK) return rc;
    AnalysisClArgs a = {};
    a.src = d_out_cl; a.band0 = d_lll; a.bandh = d_hf;
    a.C = C; a.cs = channel_stride; a.n0 = t0; a.n1 = t1; a.n2 = t2;
    a.lo0 = (2 * d0 + 2 - t0) / 2; a.lo1 = (2 * d1 + 2 - t1) / 2; a.lo2 = (2 * d2 + 2 - t2) / 2;
    a.d0 = d0; a.d1 = d1; a.d2 = d2;
    for (int i = 0; i < 8; ++i) a.taps[i] = taps[i];
    const int cw = channel_stride % 16 == 0 ? 16 : 8;
    // cells per workgroup: 64 on a large plane (256-byte runs per (channel, band): one whole line + two shared ones per
    // store instead of two shared ones: 188 -> 176 us at d = 65; 128 cells: no further gain), 32 on a small one (d = 33:
    // 23.6 vs 24.2 us, more workgroups)
    int ng = (long long)d1 * d2 >= 3072 ? 2 : 1;
    if (const char* e = getenv("LFGC_CL_ADJ_NG")) { const int v = atoi(e); if (v == 1 || v == 2) ng = v; }   // diagnostics
    const long long ptiles = ((long long)d1 * d2 + kCells * ng - 1) / (kCells * ng);
    a.ngroups = channel_stride / cw;
    if (ptiles * a.ngroups > 0x0fffffffLL) return LFGC_E_UNSUPPORTED;
    a.ptiles = (int)ptiles;
    a.zchunk = pick_zchunk(ptiles * a.ngroups, d0, (cw == 16 ? (ng == 2 ? 2 : 3) : (ng == 2 ? 4 : 6)));   // LDS 33 KB x ng per 16 channels; <= 6 waves per SIMD
    a.nchunks = (d0 + a.zchunk - 1) / a.zchunk;
    hipStream_t st = (hipStream_t)stream;
    const int lds = 2 * cw * (8 * kCells * ng + 1) * 4;
    if (ng == 2 && cw == 16) {
        static bool raised[LFGC_MAX_DEVICES] = {false};     // 65.7 KB of LDS: above the 64 KB default limit
        const int dev = lfgc_current_device();
        if (!raised[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(analysis_cl_kernel<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            raised[dev] = true;
        }
    }
    if (ng == 2) return cw == 16 ? launch_cl(analysis_cl_kernel<16, 2>, 512, a, lds, st) : launch_cl(analysis_cl_kernel<8, 2>, 256, a, lds, st);
    return cw == 16 ? launch_cl(analysis_cl_kernel<16, 1>, 512, a, lds, st) : launch_cl(analysis_cl_kernel<8, 1>, 256, a, lds, st);
}
