// lfgc_misc.hip -- ground-truth sampler, deviation statistics, parameter packing (gfx950).
#include "lfgc_common.h"
#include <math.h>

namespace {

struct GtArgs {
    const float* p; const float* f; float* out; long long n;
    int X, Y, Z;
    float min0, min1, min2, max0, max1, max2, res0, res1, res2;
};

// data/Interpolation.py:8-44, same operation order, no contraction (file is built with -ffp-contract=off
// and the arithmetic below uses the explicit round-to-nearest intrinsics).
__device__ __forceinline__ float gt_value(const GtArgs& a, long long i) {
    const float px = a.p[3 * i + 0], py = a.p[3 * i + 1], pz = a.p[3 * i + 2];
    // normalized_p = ((p - min_bb) / (max_bb - min_bb)) * (res - 1)
    const float nx = __fmul_rn(__fdiv_rn(__fsub_rn(px, a.min0), __fsub_rn(a.max0, a.min0)), __fsub_rn(a.res0, 1.0f));
    const float ny = __fmul_rn(__fdiv_rn(__fsub_rn(py, a.min1), __fsub_rn(a.max1, a.min1)), __fsub_rn(a.res1, 1.0f));
    const float nz = __fmul_rn(__fdiv_rn(__fsub_rn(pz, a.min2), __fsub_rn(a.max2, a.min2)), __fsub_rn(a.res2, 1.0f));
    const float flx = floorf(nx), fly = floorf(ny), flz = floorf(nz);
    const float cex = ceilf(nx), cey = ceilf(ny), cez = ceilf(nz);
    long long x0 = (long long)flx, y0 = (long long)fly, z0 = (long long)flz;
    long long x1 = (long long)cex, y1 = (long long)cey, z1 = (long long)cez;
    // alpha = (normalized_p - floor) / max(ceil - floor, 1e-12)  in fp64, then cast to fp32
    const double min_ref = (double)1e-12f;        // 1e-12 * ones (fp32 tensor) -> .to(double)
    const float ax = (float)(((double)nx - (double)x0) / fmax((double)(x1 - x0), min_ref));
    const float ay = (float)(((double)ny - (double)y0) / fmax((double)(y1 - y0), min_ref));
    const float az = (float)(((double)nz - (double)z0) / fmax((double)(z1 - z0), min_ref));
    const float bx = __fsub_rn(1.0f, ax), by = __fsub_rn(1.0f, ay), bz = __fsub_rn(1.0f, az);
    // torch advanced indexing wraps negative indices and raises on out-of-range; positions here are
    // lattice coordinates inside the volume (data/IndexDataset.py:91), so indices are clamped defensively.
    auto cl = [](long long v, int n) { return (int)(v < 0 ? 0 : (v >= n ? n - 1 : v)); };
    const int X0 = cl(x0, a.X), X1 = cl(x1, a.X), Y0 = cl(y0, a.Y), Y1 = cl(y1, a.Y), Z0 = cl(z0, a.Z), Z1 = cl(z1, a.Z);
    auto F = [&](int x, int y, int z) { return a.f[((long long)x * a.Y + y) * a.Z + z]; };
    const float x_y0z0 = __fadd_rn(__fmul_rn(bx, F(X0, Y0, Z0)), __fmul_rn(ax, F(X1, Y0, Z0)));
    const float x_y1z0 = __fadd_rn(__fmul_rn(bx, F(X0, Y1, Z0)), __fmul_rn(ax, F(X1, Y1, Z0)));
    const float x_y0z1 = __fadd_rn(__fmul_rn(bx, F(X0, Y0, Z1)), __fmul_rn(ax, F(X1, Y0, Z1)));
    const float x_y1z1 = __fadd_rn(__fmul_rn(bx, F(X0, Y1, Z1)), __fmul_rn(ax, F(X1, Y1, Z1)));
    const float y_z0 = __fadd_rn(__fmul_rn(by, x_y0z0), __fmul_rn(ay, x_y1z0));
    const float y_z1 = __fadd_rn(__fmul_rn(by, x_y0z1), __fmul_rn(ay, x_y1z1));
    return __fadd_rn(__fmul_rn(bz, y_z0), __fmul_rn(az, y_z1));
}

__global__ __launch_bounds__(256) void gt_interp_kernel(const GtArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.out[i] = gt_value(a, i);
}

// Ground truth + MSE of a train step in one pass (training/training.py:107-109 + :127 with nn.MSELoss): gt as above,
// diff = pred - gt, d_pred = (2/N) diff (the gradient of the mean), per-workgroup fp64 partial of sum diff^2; a second
// one-workgroup kernel folds the partials in a fixed order into loss = sum / N.
__global__ __launch_bounds__(256) void gt_mse_kernel(const GtArgs a, const float* __restrict__ pred, float* __restrict__ d_pred,
                                                     double* __restrict__ partials, float two_over_n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    double sq = 0.0;
    if (i < a.n) {
        const float g = gt_value(a, i);
        const float d = __fsub_rn(pred[i], g);
        if (a.out) a.out[i] = g;
        d_pred[i] = __fmul_rn(d, two_over_n);
        sq = (double)d * (double)d;
    }
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void mse_fold_kernel(const double* __restrict__ partials, int nparts, double inv_n,
                                                       float* __restrict__ loss) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < nparts; b += 256) acc += partials[b];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)(((s[0] + s[1]) + (s[2] + s[3])) * inv_n);
}

__device__ __forceinline__ void atomic_min_double(double* addr, double v) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *p;
    while (__longlong_as_double((long long)old) > v) {
        const unsigned long long assumed = old;
        old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}
__device__ __forceinline__ void atomic_max_double(double* addr, double v) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *p;
    while (__longlong_as_double((long long)old) < v) {
        const unsigned long long assumed = old;
        old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}

// IndexDataset.__getitem__ (data/IndexDataset.py:90-96) for flat voxel indices drawn on the device: the row of the
// (n_voxels, 3) index table (:56-57: integer lattice coordinates as fp32) and its normalisation
// scales * ((maxN - minN) * ((raw - min) / (max - min)) + minN) with maxN = 1, minN = -1 (:7-8, :95), same fp32 operations.
struct LatticeArgs {
    const long long* flat; float* raw; float* norm; long long n;
    int Y, Z;
    float min0, min1, min2, max0, max1, max2, sc0, sc1, sc2;
};

__global__ __launch_bounds__(256) void lattice_positions_kernel(const LatticeArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const long long f = a.flat[i];
    const long long yz = (long long)a.Y * a.Z;
    const float r0 = (float)(f / yz), r1 = (float)((f / a.Z) % a.Y), r2 = (float)(f % a.Z);
    auto nrm = [](float r, float mn, float mx, float sc) {
        const float q = __fdiv_rn(__fsub_rn(r, mn), __fsub_rn(mx, mn));
        return __fmul_rn(sc, __fadd_rn(__fmul_rn(2.0f, q), -1.0f));
    };
    a.raw[3 * i + 0] = r0; a.raw[3 * i + 1] = r1; a.raw[3 * i + 2] = r2;
    a.norm[3 * i + 0] = nrm(r0, a.min0, a.max0, a.sc0);
    a.norm[3 * i + 1] = nrm(r1, a.min1, a.max1, a.sc1);
    a.norm[3 * i + 2] = nrm(r2, a.min2, a.max2, a.sc2);
}

// visualization/OutputToVTK.py:53-60 partial sums (fp64 accumulation).
__global__ __launch_bounds__(256) void deviation_kernel(const float* pred, const float* gt, long long n, double* acc) {
    double sq = 0.0, ab = 0.0, mn = INFINITY, mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float g = gt[i];
        const float d = __fsub_rn(g, pred[i]);
        sq += (double)d * (double)d;
        ab += fabs((double)d);
        mn = fmin(mn, (double)g);
        mx = fmax(mx, (double)g);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sq += __shfl_down(sq, off);
        ab += __shfl_down(ab, off);
        mn = fmin(mn, __shfl_down(mn, off));
        mx = fmax(mx, __shfl_down(mx, off));
    }
    __shared__ double s[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s[wave][0] = sq; s[wave][1] = ab; s[wave][2] = mn; s[wave][3] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { sq += s[w][0]; ab += s[w][1]; mn = fmin(mn, s[w][2]); mx = fmax(mx, s[w][3]); }
        atomicAdd(&acc[0], sq);
        atomicAdd(&acc[1], ab);
        atomic_min_double(&acc[2], mn);
        atomic_max_double(&acc[3], mx);
    }
}

__global__ __launch_bounds__(256) void debug_trig_kernel(const float* x, long long n, float* so, float* co, float* sn) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float v = x[i < n ? i : n - 1];
    float s, c, k;
    lfgc_sincosf_t<false>(v, s, c);
    k = lfgc_snake_t<false>(v);
    if (__any(lfgc_trig_out_of_range(v))) {
        lfgc_sincosf_t<true>(v, s, c);
        k = lfgc_snake_t<true>(v);
    }
    if (i < n) { so[i] = s; co[i] = c; sn[i] = k; }
}

// hardware v_sin_f32 (input in revolutions) behind a v_fract range reduction -- measured only, not used: see DESIGN.md
__global__ __launch_bounds__(256) void debug_hwsin_kernel(const float* x, long long n, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x[i] * 0.15915494309189535f));
}

struct PackArgs {
    const float* w[LFGC_MAX_LAYERS + 1];
    const float* b[LFGC_MAX_LAYERS + 1];
    float* packed;
    LfgcPlan plan;
};

// Per-layer power-of-two scales for the f16-split images: 2^S with max|w| * 2^S in [2^13, 2^14), so that the hi
// halves stay far below the f16 maximum and the lo halves (<= 2^-11 of the hi) stay normal for every weight within
// 2^-16 of the largest.  Two sets: the transposed images of the backward chain hold W itself, the forward images hold
// W / pi (layer 0) or W / (pi LFGC_ACT_SCALE) (lfgc_forward16.h: pre-activations in turns of pi, inputs of layers >= 1
// scaled).  One block per hidden layer.
__device__ __forceinline__ double lfgc_fwd_image_divisor(int l) {
    return l == 0 ? 3.14159265358979323846 : 3.14159265358979323846 * LFGC_ACT_SCALE;
}

// The draw of a train step and its positions in ONE kernel (the data/IndexDataset.py:90-96 sampler with the indices
// drawn on the device): flat index i of draw number `step` = floor(u64 * n_voxels / 2^64) with
// u64 = the first two words of Philox4x32-10(counter = (i, step), key = seed) -- counter-based, so a replayed HIP graph
// draws a new batch every replay: `state[0]` (device) holds the step; the last workgroup to finish advances it
// (every workgroup has read it by then: ticket in state[1], left at 0).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned* r0, unsigned* r1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    *r0 = c0; *r1 = c1;
}

struct LatticeSampleArgs {
    LatticeArgs l;                 // l.flat unused
    long long* flat_out;           // optional
    unsigned long long* state;     // [0] step, [1] ticket
    unsigned long long seed, n_voxels;
};

__global__ __launch_bounds__(256) void lattice_sample_kernel(const LatticeSampleArgs a) {
    const unsigned long long step = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < a.l.n) {
        unsigned r0, r1;
        philox4x32_10((unsigned)i, (unsigned)((unsigned long long)i >> 32), (unsigned)step, (unsigned)(step >> 32),
                      (unsigned)a.seed, (unsigned)(a.seed >> 32), &r0, &r1);
        const long long f = (long long)__umul64hi(((unsigned long long)r0 << 32) | r1, a.n_voxels);
        if (a.flat_out) a.flat_out[i] = f;
        const long long yz = (long long)a.l.Y * a.l.Z;
        const float q0 = (float)(f / yz), q1 = (float)((f / a.l.Z) % a.l.Y), q2 = (float)(f % a.l.Z);
        auto nrm = [](float r, float mn, float mx, float sc) {
            const float q = __fdiv_rn(__fsub_rn(r, mn), __fsub_rn(mx, mn));
            return __fmul_rn(sc, __fadd_rn(__fmul_rn(2.0f, q), -1.0f));
        };
        a.l.raw[3 * i + 0] = q0; a.l.raw[3 * i + 1] = q1; a.l.raw[3 * i + 2] = q2;
        a.l.norm[3 * i + 0] = nrm(q0, a.l.min0, a.l.max0, a.l.sc0);
        a.l.norm[3 * i + 1] = nrm(q1, a.l.min1, a.l.max1, a.l.sc1);
        a.l.norm[3 * i + 2] = nrm(q2, a.l.min2, a.l.max2, a.l.sc2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = __hip_atomic_fetch_add(a.state + 1, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(a.state + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.state, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ int lfgc_pow2_exponent_for(float m) {
    int e = 0;
    if (m > 0.0f && m < INFINITY) { (void)frexpf(m, &e); e = 14 - e; }    // m = f * 2^e', f in [0.5,1) -> m * 2^(14-e') in [2^13, 2^14)
    if (e > 60) e = 60;
    if (e < -60) e = -60;
    return e;
}

__global__ __launch_bounds__(1024) void pack_scale_kernel(const PackArgs a) {
    const LfgcPlan& p = a.plan;
    const int l = blockIdx.x;
    const int n = (l == 0) ? p.H * (p.E + p.C) : p.H * p.H;
    const float* w = a.w[l];
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;      // 4 loads in flight per pass (it runs in every training forward)
    for (int i = threadIdx.x; i < n; i += 4096) {
        m0 = fmaxf(m0, fabsf(w[i]));
        if (i + 1024 < n) m1 = fmaxf(m1, fabsf(w[i + 1024]));
        if (i + 2048 < n) m2 = fmaxf(m2, fabsf(w[i + 2048]));
        if (i + 3072 < n) m3 = fmaxf(m3, fabsf(w[i + 3072]));
    }
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
    __shared__ float s[16];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = 0.0f;
        for (int k = 0; k < 16; ++k) m = fmaxf(m, s[k]);
        const int e = lfgc_pow2_exponent_for(m);
        a.packed[p.off_h + l] = ldexpf(1.0f, e);
        a.packed[p.off_h + 8 + l] = ldexpf(1.0f, -e);
        const int ef = lfgc_pow2_exponent_for((float)((double)m / lfgc_fwd_image_divisor(l)));
        a.packed[p.off_h + 16 + l] = ldexpf(1.0f, ef);
        a.packed[p.off_h + 24 + l] = ldexpf(1.0f, -ef);
    }
}

// Re-lay nn.Linear parameters into the blob described in lfgc_common.h.
__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a) {
    const LfgcPlan& p = a.plan;
    const int K0 = p.E + p.C;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < p.total_floats; idx += gridDim.x * 256) {
        float v = 0.0f;
        if (idx < p.blk0) {                                   // layer 0: permuted columns
            if (idx < p.HP * p.S0) {
                const int row = idx / p.S0, cl = idx % p.S0;
                if (row < p.H && cl < p.K0P) {
                    const int src = lfgc_layer0_src_col(p, cl);
                    if (src >= 0) v = a.w[0][row * K0 + src];
                }
            } else {
                const int r = idx - p.HP * p.S0;
                if (r < p.H) v = a.b[0][r];
            }
        } else if (idx < p.off_final) {                       // hidden layers: natural columns
            const int l = 1 + (idx - p.blk0) / p.blk1;
            const int o = (idx - p.blk0) % p.blk1;
            if (o < p.HP * p.S1) {
                const int row = o / p.S1, cl = o % p.S1;
                if (row < p.H && cl < p.H) v = a.w[l][row * p.H + cl];
            } else {
                const int r = o - p.HP * p.S1;
                if (r < p.H) v = a.b[l][r];
            }
        } else if (idx < p.fwd_floats) {                      // final layer
            const int o = idx - p.off_final;
            if (o < p.H) v = a.w[p.L][o];
            else if (o == p.HP) v = a.b[p.L][0];
        } else if (idx < p.off_t + p.tblk0) {                 // layer 0 transposed: [packed col][h_out]
            const int o = idx - p.off_t;
            const int cl = o / p.ST, ho = o % p.ST;
            if (ho < p.H && cl < p.K0P) {
                const int src = lfgc_layer0_src_col(p, cl);
                if (src >= 0) v = a.w[0][ho * K0 + src];
            }
        } else if (idx < p.off_h) {                           // hidden layers transposed: [k_in][h_out]
            const int o = idx - p.off_t - p.tblk0;
            const int l = 1 + o / p.tblk1;
            const int oo = o % p.tblk1;
            const int ki = oo / p.ST, ho = oo % p.ST;
            if (ki < p.H && ho < p.H) v = a.w[l][ho * p.H + ki];
        } else if (idx < p.off_hbias) {
            continue;                                         // scales: written by pack_scale_kernel
        } else if (idx < p.off_hwf) {                         // b / pi of the hidden layers, [layer][row]
            const int l = (idx - p.off_hbias) / p.HP, r = (idx - p.off_hbias) % p.HP;
            if (l < p.L && r < p.H) v = (float)((double)a.b[l][r] / 3.14159265358979323846);
        } else if (idx < p.off_hblk) {                        // head weights / LFGC_ACT_SCALE
            const int r = idx - p.off_hwf;
            if (r < p.H) v = (float)((double)a.w[p.L][r] / LFGC_ACT_SCALE);
        } else if (idx >= p.off_ht) {                         // f16-split transposed images: [row = k_in][k = h_out]
            const int o = idx - p.off_ht;
            const int l = o < p.tblk0 ? 0 : 1 + (o - p.tblk0) / p.tblk1;
            const int oo = o < p.tblk0 ? o : (o - p.tblk0) % p.tblk1;
            const int row = oo / p.ST, q = oo % p.ST;
            const float scale = a.packed[p.off_h + l];
            unsigned bits = 0;
            const int in_col = (l == 0) ? (row < p.K0P ? lfgc_layer0_src_col(p, row) : -1) : (row < p.H ? row : -1);
            if (in_col >= 0 && q < p.HP) {
                const int b = q >> 4, t = q & 15;
                const int hp = t >> 3, part = (t >> 2) & 1, pair = t & 3;
                const int Kin = l == 0 ? (p.E + p.C) : p.H;
                unsigned short hv[2] = {0, 0};
                for (int u = 0; u < 2; ++u) {
                    const int jj = 2 * pair + u;
                    const int ho = 16 * b + (jj & 3) + 8 * (jj >> 2) + 4 * hp;       // contraction index = h_out
                    if (ho < p.H) {
                        const float w = a.w[l][ho * Kin + in_col] * scale;
                        const _Float16 hi = (_Float16)w;
                        const _Float16 lo = (_Float16)(w - (float)hi);
                        const _Float16 x = part ? lo : hi;
                        hv[u] = *reinterpret_cast<const unsigned short*>(&x);
                    }
                }
                bits = (unsigned)hv[0] | ((unsigned)hv[1] << 16);
            }
            v = __uint_as_float(bits);
        } else {                                              // f16-split blocks (hi | lo halves, scaled)
            const int o = idx - p.off_hblk;
            const int l = o < p.blkh0 ? 0 : 1 + (o - p.blkh0) / p.blkh1;
            const int oo = o < p.blkh0 ? o : (o - p.blkh0) % p.blkh1;
            const int SH = l == 0 ? p.SH0 : p.SH1;
            const int K = l == 0 ? p.K0P16 : p.HP;
            const int Kin = l == 0 ? (p.E + p.C) : p.H;
            // W / pi [/ LFGC_ACT_SCALE] in fp64, then the power-of-two scale, then the split: the image carries the
            // quotient to the 22-24 bits of the split itself
            const double scale = (double)a.packed[p.off_h + 16 + l] / lfgc_fwd_image_divisor(l);
            if (oo < p.HP * SH) {
                const int row = oo / SH, q = oo % SH;
                unsigned bits = 0;
                if (row < p.H && q < K) {
                    const int b = q >> 4, t = q & 15;
                    const int hp = t >> 3, part = (t >> 2) & 1, pair = t & 3;
                    unsigned short hv[2] = {0, 0};
                    for (int u = 0; u < 2; ++u) {
                        const int src = lfgc_h16_src_col(p, l, b, hp, 2 * pair + u);
                        if (src >= 0) {
                            const double w = (double)a.w[l][row * Kin + src] * scale;
                            const _Float16 hi = (_Float16)w;
                            const _Float16 lo = (_Float16)(w - (double)hi);
                            const _Float16 x = part ? lo : hi;
                            hv[u] = *reinterpret_cast<const unsigned short*>(&x);
                        }
                    }
                    bits = (unsigned)hv[0] | ((unsigned)hv[1] << 16);
                }
                v = __uint_as_float(bits);
            } else {
                const int r = oo - p.HP * SH;
                v = 0.0f;                                     // (bias slot of the block: unused, the biases live at off_hbias)
            }
        }
        a.packed[idx] = v;
    }
}

}  // namespace

extern "C" int lfgc_gt_interp_f32(const float* p, const float* f, const float* min_bb, const float* max_bb,
                                  const float* res, int64_t n, int X, int Y, int Z, float* out, lfgc_stream_t stream) {
    if (!p || !f || !min_bb || !max_bb || !res || !out) return LFGC_E_NULL;
    if (n < 0 || X < 1 || Y < 1 || Z < 1) return LFGC_E_SHAPE;
    if (n == 0) return LFGC_OK;
    GtArgs a;
    a.p = p; a.f = f; a.out = out; a.n = n; a.X = X; a.Y = Y; a.Z = Z;
    a.min0 = min_bb[0]; a.min1 = min_bb[1]; a.min2 = min_bb[2];
    a.max0 = max_bb[0]; a.max1 = max_bb[1]; a.max2 = max_bb[2];
    a.res0 = res[0]; a.res1 = res[1]; a.res2 = res[2];
    hipLaunchKernelGGL(gt_interp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int64_t lfgc_gt_mse_workspace_bytes(int64_t n) { return n < 1 ? 8 : ((n + 255) / 256) * 8; }

extern "C" int lfgc_gt_mse_f32(const float* p, const float* f, const float* min_bb, const float* max_bb, const float* res,
                               int64_t n, int X, int Y, int Z, const float* pred, float* gt_out, float* d_pred, float* loss,
                               void* workspace, int64_t workspace_bytes, lfgc_stream_t stream) {
    if (!p || !f || !min_bb || !max_bb || !res || !pred || !d_pred || !loss || !workspace) return LFGC_E_NULL;
    if (n < 1 || X < 1 || Y < 1 || Z < 1) return LFGC_E_SHAPE;
    if (workspace_bytes < lfgc_gt_mse_workspace_bytes(n)) return LFGC_E_WORKSPACE;
    GtArgs a;
    a.p = p; a.f = f; a.out = gt_out; a.n = n; a.X = X; a.Y = Y; a.Z = Z;
    a.min0 = min_bb[0]; a.min1 = min_bb[1]; a.min2 = min_bb[2];
    a.max0 = max_bb[0]; a.max1 = max_bb[1]; a.max2 = max_bb[2];
    a.res0 = res[0]; a.res1 = res[1]; a.res2 = res[2];
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return LFGC_E_UNSUPPORTED;
    double* partials = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(gt_mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, pred, d_pred, partials,
                       (float)(2.0 / (double)n));
    LFGC_HIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)blocks, 1.0 / (double)n, loss);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_lattice_positions_f32(const int64_t* flat, int64_t n, const int32_t* res, const float* min_idx,
                                          const float* max_idx, const float* scales, float* raw, float* norm,
                                          lfgc_stream_t stream) {
    if (!flat || !res || !min_idx || !max_idx || !scales || !raw || !norm) return LFGC_E_NULL;
    if (n < 0 || res[0] < 1 || res[1] < 1 || res[2] < 1) return LFGC_E_SHAPE;
    if (n == 0) return LFGC_OK;
    LatticeArgs a;
    a.flat = reinterpret_cast<const long long*>(flat); a.raw = raw; a.norm = norm; a.n = n;
    a.Y = res[1]; a.Z = res[2];
    a.min0 = min_idx[0]; a.min1 = min_idx[1]; a.min2 = min_idx[2];
    a.max0 = max_idx[0]; a.max1 = max_idx[1]; a.max2 = max_idx[2];
    a.sc0 = scales[0]; a.sc1 = scales[1]; a.sc2 = scales[2];
    hipLaunchKernelGGL(lattice_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_lattice_sample_f32(uint64_t seed, int64_t* state, int64_t n, const int32_t* res, const float* min_idx,
                                       const float* max_idx, const float* scales, float* raw, float* norm, int64_t* flat_out,
                                       lfgc_stream_t stream) {
    if (!state || !res || !min_idx || !max_idx || !scales || !raw || !norm) return LFGC_E_NULL;
    if (n < 0 || res[0] < 1 || res[1] < 1 || res[2] < 1) return LFGC_E_SHAPE;
    if (n == 0) return LFGC_OK;
    LatticeSampleArgs a;
    a.l.flat = nullptr; a.l.raw = raw; a.l.norm = norm; a.l.n = n;
    a.l.Y = res[1]; a.l.Z = res[2];
    a.l.min0 = min_idx[0]; a.l.min1 = min_idx[1]; a.l.min2 = min_idx[2];
    a.l.max0 = max_idx[0]; a.l.max1 = max_idx[1]; a.l.max2 = max_idx[2];
    a.l.sc0 = scales[0]; a.l.sc1 = scales[1]; a.l.sc2 = scales[2];
    a.flat_out = reinterpret_cast<long long*>(flat_out);
    a.state = reinterpret_cast<unsigned long long*>(state);
    a.seed = seed; a.n_voxels = (unsigned long long)res[0] * (unsigned long long)res[1] * (unsigned long long)res[2];
    hipLaunchKernelGGL(lattice_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_deviation_partial_f32(const float* pred, const float* gt, int64_t n, double* acc, lfgc_stream_t stream) {
    if (!pred || !gt || !acc) return LFGC_E_NULL;
    if (n < 0) return LFGC_E_SHAPE;
    if (n == 0) return LFGC_OK;
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(deviation_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pred, gt, (long long)n, acc);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_mlp_supported(const lfgc_mlp_desc* d) {
    if (!d) return 0;
    if (d->d_in != 3 || d->d_out != 1) return 0;
    if (d->grid_channels < 1 || d->grid_channels > 32) return 0;
    if (d->hidden < 1 || d->hidden > 128) return 0;
    if (d->num_layers < 1 || d->num_layers > LFGC_MAX_LAYERS) return 0;
    if (d->n_freqs != 2) return 0;
    return 1;
}

extern "C" int lfgc_grid_channel_stride(int C) { return lfgc_roundup(C, 8); }

extern "C" int64_t lfgc_packed_bytes(const lfgc_mlp_desc* d) {
    if (!lfgc_mlp_supported(d)) return LFGC_E_UNSUPPORTED;
    return (int64_t)lfgc_make_plan(d->grid_channels, d->hidden, d->num_layers, d->n_freqs).total_floats * 4;
}

extern "C" int64_t lfgc_stash_bytes(const lfgc_mlp_desc* d, int64_t n) {
    if (!lfgc_mlp_supported(d)) return LFGC_E_UNSUPPORTED;
    if (n < 0) return LFGC_E_SHAPE;
    const LfgcPlan p = lfgc_make_plan(d->grid_channels, d->hidden, d->num_layers, d->n_freqs);
    const int64_t tiles = (n + 255) / 256 * 8;      // whole 256-sample workgroup batches of 32-sample tiles
    return tiles * (int64_t)p.stash_tile_floats * 4;
}

extern "C" int lfgc_pack_mlp_f32(const lfgc_mlp_desc* d, const float* const* weights, const float* const* biases,
                                 float* packed, lfgc_stream_t stream) {
    if (!d || !weights || !biases || !packed) return LFGC_E_NULL;
    if (!lfgc_mlp_supported(d)) return LFGC_E_UNSUPPORTED;
    if (((uintptr_t)packed) & 15) return LFGC_E_ALIGN;
    PackArgs a;
    a.plan = lfgc_make_plan(d->grid_channels, d->hidden, d->num_layers, d->n_freqs);
    for (int l = 0; l <= d->num_layers; ++l) {
        if (!weights[l] || !biases[l]) return LFGC_E_NULL;
        a.w[l] = weights[l];
        a.b[l] = biases[l];
    }
    a.packed = packed;
    hipLaunchKernelGGL(pack_scale_kernel, dim3(a.plan.L), dim3(1024), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    const int g = (a.plan.total_floats + 255) / 256;
    hipLaunchKernelGGL(pack_kernel, dim3(g > 1024 ? 1024 : g), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_debug_trig_f32(const float* x, int64_t n, float* sin_out, float* cos_out, float* snake_out,
                                   lfgc_stream_t stream) {
    if (!x || !sin_out || !cos_out || !snake_out) return LFGC_E_NULL;
    if (n <= 0) return n == 0 ? LFGC_OK : LFGC_E_SHAPE;
    hipLaunchKernelGGL(debug_trig_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, (long long)n, sin_out, cos_out, snake_out);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_debug_hwsin_f32(const float* x, int64_t n, float* out, lfgc_stream_t stream) {
    if (!x || !out) return LFGC_E_NULL;
    if (n <= 0) return n == 0 ? LFGC_OK : LFGC_E_SHAPE;
    hipLaunchKernelGGL(debug_hwsin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, out);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_version(void) { return LFGC_VERSION; }

extern "C" const char* lfgc_error_string(int code) {
    switch (code) {
        case LFGC_OK: return "ok";
        case LFGC_E_NULL: return "required pointer is NULL";
        case LFGC_E_SHAPE: return "invalid or inconsistent extent";
        case LFGC_E_UNSUPPORTED: return "network shape outside the compiled kernel set";
        case LFGC_E_ALIGN: return "pointer must be 16-byte aligned";
        case LFGC_E_WORKSPACE: return "workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown lfgc error";
    }
}
