// lfgc_drop.hip -- the pruning ("drop") layers around the wavelet coefficients, outside the fused IDWT (gfx950).
//   lfgc_drop_apply_f32 / _bwd_f32      one drop layer on one coefficient tensor (the layers' own forward(x):
//                                       model/Smallify_Dropout.py:54-61, model/Straight_Through_Dropout.py:26-30, :54-62,
//                                       model/Variational_Dropout_Layer.py:101-112); the fused form lives in lfgc_wavelet.hip
//   lfgc_sign_variance_update_f32       SmallifySignVarianceTracker.sign_variance_pruning_onlyVar (Smallify_Dropout.py:106-112)
//                                       on device state (the reference keeps it on the CPU and syncs every step)
//   lfgc_penalty_sums_f32 / _grads_f32  L1 of the drop parameters, sum G^2 of the coefficients, D_KL of the variational
//                                       layers (Smallify_Dropout.py:21-40, :63-64; Variational_Dropout_Layer.py:48-53, :115-122)
//                                       as ONE multi-tensor reduction and ONE multi-tensor gradient kernel
// All byte movers / reductions: coalesced rows, fp64 accumulators, nothing worth an MFMA.
#include "lfgc_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float drop_value(float x, float m, float thr, bool ste) {
    if (!ste) return __fmul_rn(x, m);
    const float hard = m >= thr ? 1.0f : 0.0f;
    const float soft = __fmul_rn(x, m);
    return __fadd_rn(__fsub_rn(__fmul_rn(x, hard), soft), soft);
}

__global__ __launch_bounds__(256) void drop_apply_kernel(const float* __restrict__ x, const float* __restrict__ mul,
                                                         float thr, float* __restrict__ out, long long total, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = drop_value(x[i], mul[i % n], thr, thr == thr);
}

// d_x[c][i] = g[c][i] * m[i];  d_m[i] = sum_c g[c][i] * x[c][i]   (thread = cell, lanes along cells, fixed channel order)
__global__ __launch_bounds__(256) void drop_apply_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                             const float* __restrict__ mul, float* __restrict__ d_x,
                                                             float* __restrict__ d_mul, int C, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float m = mul[i];
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float gv = g[(long long)c * n + i];
        if (d_mul) acc = __builtin_fmaf(gv, x[(long long)c * n + i], acc);
        d_x[(long long)c * n + i] = gv * m;
    }
    if (d_mul) d_mul[i] = acc;
}

// phi = sign(beta) - EMA;  EMA += mom * phi;  EMAVar = (1 - mom) * (EMAVar + mom * phi^2)      -- the reference's fp32
// operation order (Smallify_Dropout.py:108-112), so the state is bit-identical to its CPU tensors.
__global__ __launch_bounds__(256) void sign_variance_kernel(const float* __restrict__ betas, float* __restrict__ ema,
                                                            float* __restrict__ emavar, float mom, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float b = betas[i];
    const float sgn = b != b ? b : (float)((b > 0.0f) - (b < 0.0f));
    const float e = ema[i];
    const float phi = __fsub_rn(sgn, e);
    ema[i] = __fadd_rn(e, __fmul_rn(mom, phi));
    emavar[i] = __fmul_rn(__fsub_rn(1.0f, mom), __fadd_rn(emavar[i], __fmul_rn(mom, __fmul_rn(phi, phi))));
}

struct TrackerArgs {
    const float* betas[LFGC_PENALTY_MAX_TERMS];
    float* ema[LFGC_PENALTY_MAX_TERMS];
    float* emavar[LFGC_PENALTY_MAX_TERMS];
    long long n[LFGC_PENALTY_MAX_TERMS];
    int block_start[LFGC_PENALTY_MAX_TERMS + 1];
    int n_layers;
    float mom;
};

__global__ __launch_bounds__(256) void sign_variance_multi_kernel(const TrackerArgs a) {
    int t = 0;
    while (t + 1 < a.n_layers && (int)blockIdx.x >= a.block_start[t + 1]) ++t;
    const long long i = (long long)(blockIdx.x - a.block_start[t]) * 256 + threadIdx.x;
    if (i >= a.n[t]) return;
    const float b = a.betas[t][i];
    const float sgn = b != b ? b : (float)((b > 0.0f) - (b < 0.0f));
    const float e = a.ema[t][i];
    const float phi = __fsub_rn(sgn, e);
    a.ema[t][i] = __fadd_rn(e, __fmul_rn(a.mom, phi));
    a.emavar[t][i] = __fmul_rn(__fsub_rn(1.0f, a.mom), __fadd_rn(a.emavar[t][i], __fmul_rn(a.mom, __fmul_rn(phi, phi))));
}

constexpr int kMaxTerms = LFGC_PENALTY_MAX_TERMS;
struct PenaltyArgs {
    lfgc_penalty_term term[kMaxTerms];
    float* grad_a[kMaxTerms];
    float* grad_b[kMaxTerms];
    int block_start[kMaxTerms + 1];   // 1-D grid: blocks [block_start[t], block_start[t+1]) work on term t
    int n_terms;
};

__device__ __forceinline__ int term_of_block(const PenaltyArgs& a, int block) {
    int t = 0;
    while (t + 1 < a.n_terms && block >= a.block_start[t + 1]) ++t;
    return t;
}

// Molchanov et al. constants as the reference multiplies them into fp32 tensors (Variational_Dropout_Layer.py:74-77)
__device__ __forceinline__ double kl_k1() { return (double)0.63576f; }
__device__ __forceinline__ double kl_k2() { return (double)1.87320f; }
__device__ __forceinline__ double kl_k3() { return (double)1.48695f; }

__device__ __forceinline__ double penalty_value(int kind, float a, float b) {
    if (kind == LFGC_PENALTY_L1) return fabs((double)a);
    if (kind == LFGC_PENALTY_L2) return (double)a * (double)a;
    const double la = (double)b - 2.0 * (double)a;                       // log alpha = log_var - 2 log_theta
    const double t1 = kl_k1() / (1.0 + exp(-(kl_k2() + kl_k3() * la)));
    const double t2 = 0.5 * (la < 0.0 ? -la + log1p(exp(la)) : log1p(exp(-la)));   // 0.5 * softplus(-la)
    return -t1 + t2 + kl_k1();
}

__global__ __launch_bounds__(256) void penalty_sums_kernel(const PenaltyArgs a, double* __restrict__ sums) {
    const int t = term_of_block(a, blockIdx.x);
    const lfgc_penalty_term term = a.term[t];
    const int nb = a.block_start[t + 1] - a.block_start[t];
    const long long first = (long long)(blockIdx.x - a.block_start[t]) * 1024 + threadIdx.x, stride = (long long)nb * 1024;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    const bool dkl = term.kind == LFGC_PENALTY_DKL;
    for (long long i = first; i < term.n; i += stride) {  // 4 independent loads / accumulators per pass
        const long long i1 = i + 256, i2 = i + 512, i3 = i + 768;
        const float a0 = term.a[i], a1 = i1 < term.n ? term.a[i1] : 0.0f, a2 = i2 < term.n ? term.a[i2] : 0.0f,
                    a3 = i3 < term.n ? term.a[i3] : 0.0f;
        if (!dkl) {
            acc0 += penalty_value(term.kind, a0, 0.0f); acc1 += penalty_value(term.kind, a1, 0.0f);
            acc2 += penalty_value(term.kind, a2, 0.0f); acc3 += penalty_value(term.kind, a3, 0.0f);
        } else {
            acc0 += penalty_value(LFGC_PENALTY_DKL, a0, term.b[i]);
            if (i1 < term.n) acc1 += penalty_value(LFGC_PENALTY_DKL, a1, term.b[i1]);
            if (i2 < term.n) acc2 += penalty_value(LFGC_PENALTY_DKL, a2, term.b[i2]);
            if (i3 < term.n) acc3 += penalty_value(LFGC_PENALTY_DKL, a3, term.b[i3]);
        }
    }
    double acc = (acc0 + acc1) + (acc2 + acc3);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    // per-block partial; a second tiny kernel folds them in a fixed order (same-address fp64 atomics from thousands of
    // blocks serialise at ~20 ns each, and a last-block fold needs device-scope fences = L2 write-backs on this chip)
    if (threadIdx.x == 0) sums[a.n_terms + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void penalty_fold_kernel(const PenaltyArgs a, double* __restrict__ sums) {
    const int t = blockIdx.x;
    double acc = 0.0;
    for (int b = a.block_start[t] + threadIdx.x; b < a.block_start[t + 1]; b += 256) acc += sums[a.n_terms + b];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[t] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void penalty_grads_kernel(const PenaltyArgs a, const float* __restrict__ d_sums) {
    const int t = term_of_block(a, blockIdx.x);
    const lfgc_penalty_term term = a.term[t];
    const float g = d_sums[t];
    float* ga = a.grad_a[t];
    float* gb = a.grad_b[t];
    const int nb = a.block_start[t + 1] - a.block_start[t];
    for (long long i = (long long)(blockIdx.x - a.block_start[t]) * 256 + threadIdx.x; i < term.n; i += (long long)nb * 256) {
        const float v = term.a[i];
        if (term.kind == LFGC_PENALTY_L1) {
            ga[i] = g * (float)((v > 0.0f) - (v < 0.0f));
        } else if (term.kind == LFGC_PENALTY_L2) {
            ga[i] = g * (2.0f * v);
        } else {
            const double la = (double)term.b[i] - 2.0 * (double)v;
            const double s = 1.0 / (1.0 + exp(-(kl_k2() + kl_k3() * la)));
            const double d_la = -kl_k1() * kl_k3() * s * (1.0 - s) - 0.5 / (1.0 + exp(la));   // d/dla of -t1 + t2
            gb[i] = (float)((double)g * d_la);
            ga[i] = (float)((double)g * (-2.0 * d_la));
        }
    }
}

inline unsigned blocks_for(long long n, long long cap) {
    long long g = (n + 255) / 256;
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int lfgc_drop_apply_f32(const float* x, const float* mul, float threshold, float* out, int C, int64_t n,
                                   lfgc_stream_t stream) {
    if (!x || !mul || !out) return LFGC_E_NULL;
    if (C < 1 || n < 1) return LFGC_E_SHAPE;
    const long long total = (long long)C * n;
    hipLaunchKernelGGL(drop_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, mul, threshold, out, total, (long long)n);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_drop_apply_bwd_f32(const float* d_out, const float* x, const float* mul, float* d_x, float* d_mul,
                                       int C, int64_t n, lfgc_stream_t stream) {
    if (!d_out || !mul || !d_x || (d_mul && !x)) return LFGC_E_NULL;
    if (C < 1 || n < 1) return LFGC_E_SHAPE;
    hipLaunchKernelGGL(drop_apply_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_out, x, mul, d_x, d_mul, C, (long long)n);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_sign_variance_update_f32(const float* betas, float* ema, float* emavar, float momentum, int64_t n,
                                             lfgc_stream_t stream) {
    if (!betas || !ema || !emavar) return LFGC_E_NULL;
    if (n < 1) return LFGC_E_SHAPE;
    hipLaunchKernelGGL(sign_variance_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       betas, ema, emavar, momentum, (long long)n);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_sign_variance_update_multi_f32(const float* const* betas, float* const* ema, float* const* emavar,
                                                   const int64_t* n, int n_layers, float momentum, lfgc_stream_t stream) {
    if (!betas || !ema || !emavar || !n) return LFGC_E_NULL;
    if (n_layers < 1 || n_layers > LFGC_PENALTY_MAX_TERMS) return LFGC_E_SHAPE;
    TrackerArgs a;
    int total = 0;
    for (int t = 0; t < n_layers; ++t) {
        if (!betas[t] || !ema[t] || !emavar[t]) return LFGC_E_NULL;
        if (n[t] < 1) return LFGC_E_SHAPE;
        a.betas[t] = betas[t]; a.ema[t] = ema[t]; a.emavar[t] = emavar[t]; a.n[t] = n[t];
        a.block_start[t] = total;
        total += (int)((n[t] + 255) / 256);
    }
    a.block_start[n_layers] = total;
    a.n_layers = n_layers; a.mom = momentum;
    hipLaunchKernelGGL(sign_variance_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

static int penalty_check(const lfgc_penalty_term* terms, int n_terms, long long* max_n) {
    if (!terms) return LFGC_E_NULL;
    if (n_terms < 1 || n_terms > LFGC_PENALTY_MAX_TERMS) return LFGC_E_SHAPE;
    *max_n = 0;
    for (int t = 0; t < n_terms; ++t) {
        if (!terms[t].a || (terms[t].kind == LFGC_PENALTY_DKL && !terms[t].b)) return LFGC_E_NULL;
        if (terms[t].n < 0 || terms[t].kind < LFGC_PENALTY_L1 || terms[t].kind > LFGC_PENALTY_DKL) return LFGC_E_SHAPE;
        if (terms[t].n > *max_n) *max_n = terms[t].n;
    }
    return LFGC_OK;
}

extern "C" int lfgc_penalty_sums_f32(const lfgc_penalty_term* terms, int n_terms, double* sums, lfgc_stream_t stream) {
    long long max_n;
    const int rc = penalty_check(terms, n_terms, &max_n);
    if (rc != LFGC_OK) return rc;
    if (!sums) return LFGC_E_NULL;
    PenaltyArgs a;
    for (int t = 0; t < n_terms; ++t) { a.term[t] = terms[t]; a.grad_a[t] = nullptr; a.grad_b[t] = nullptr; }
    a.n_terms = n_terms;
    int total = 0;
    for (int t = 0; t < n_terms; ++t) { a.block_start[t] = total; total += (int)blocks_for((terms[t].n + 3) / 4, LFGC_PENALTY_BLOCKS); }
    a.block_start[n_terms] = total;
    hipLaunchKernelGGL(penalty_sums_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a, sums);
    LFGC_HIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(penalty_fold_kernel, dim3(n_terms), dim3(256), 0, (hipStream_t)stream, a, sums);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_penalty_grads_f32(const lfgc_penalty_term* terms, int n_terms, const float* d_sums,
                                      float* const* grad_a, float* const* grad_b, lfgc_stream_t stream) {
    long long max_n;
    const int rc = penalty_check(terms, n_terms, &max_n);
    if (rc != LFGC_OK) return rc;
    if (!d_sums || !grad_a) return LFGC_E_NULL;
    PenaltyArgs a;
    for (int t = 0; t < n_terms; ++t) {
        a.term[t] = terms[t];
        a.grad_a[t] = grad_a[t];
        a.grad_b[t] = grad_b ? grad_b[t] : nullptr;
        if (!a.grad_a[t] || (terms[t].kind == LFGC_PENALTY_DKL && !a.grad_b[t])) return LFGC_E_NULL;
    }
    a.n_terms = n_terms;
    int total = 0;
    for (int t = 0; t < n_terms; ++t) { a.block_start[t] = total; total += (int)blocks_for(terms[t].n, 2048); }
    a.block_start[n_terms] = total;
    hipLaunchKernelGGL(penalty_grads_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a, d_sums);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}
