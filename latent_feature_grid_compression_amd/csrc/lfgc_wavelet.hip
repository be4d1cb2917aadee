// lfgc_wavelet.hip -- db2 (any 4-tap) separable-bank 3-D wavelet kernels for gfx950.
//   lfgc_idwt_level_f32      replaces wavelet_transform/Torch_Wavelet_Transform.py:91-104 (+ crop :69-73)
//   lfgc_idwt_level_bwd_f32  its adjoint (autograd of the same lines)
//   lfgc_dwt_level_f32       replaces :59-67, :75-89 (init-time encode)
// All three are HBM/L2-bound stencil kernels (no contraction worth an MFMA): coalesced along the
// last spatial axis, filter bank (8 x 64 taps) in LDS.
#include "lfgc_common.h"

namespace {

struct IdwtArgs {
    const float* lll;   // (C, d0,d1,d2)
    const float* hf;    // (C, 7, d0,d1,d2)
    const float* filt;  // (8,4,4,4)
    float* out;
    int C, d0, d1, d2, t0, t1, t2, o0, o1, o2;   // o = crop offset floor((2d+2-t)/2)
    int channel_last, cs;
};

// Synthesis: out_full[o] = sum_{s,t} in[s][i] F_s[t], o = 2 i + t per axis  ->  for output parity p and
// cell jj = o_full >> 1 the contributing (i, t) are (jj, p) and (jj - 1, p + 2).
__global__ __launch_bounds__(256) void idwt_level_kernel(const IdwtArgs a) {
    __shared__ float s_f[512];
    for (int i = threadIdx.x; i < 512; i += 256) s_f[i] = a.filt[i];
    __syncthreads();
    const long long total = (long long)a.C * a.t0 * a.t1 * a.t2;
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ox = (int)(idx % a.t2);
        long long r = idx / a.t2;
        const int oy = (int)(r % a.t1); r /= a.t1;
        const int oz = (int)(r % a.t0);
        const int c = (int)(r / a.t0);
        const int fz = oz + a.o0, fy = oy + a.o1, fx = ox + a.o2;
        const int pz = fz & 1, py = fy & 1, px = fx & 1;
        const int jz = fz >> 1, jy = fy >> 1, jx = fx >> 1;
        const float* in_l = a.lll + (long long)c * dvol;
        const float* in_h = a.hf + (long long)c * 7 * dvol;
        float acc = 0.0f;
#pragma unroll
        for (int ez = 0; ez < 2; ++ez) {
            const int iz = jz - ez, tz = pz + 2 * ez;
            if (iz < 0 || iz >= a.d0) continue;
#pragma unroll
            for (int ey = 0; ey < 2; ++ey) {
                const int iy = jy - ey, ty = py + 2 * ey;
                if (iy < 0 || iy >= a.d1) continue;
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    const int ix = jx - ex, tx = px + 2 * ex;
                    if (ix < 0 || ix >= a.d2) continue;
                    const long long sp = ((long long)iz * a.d1 + iy) * a.d2 + ix;
                    const int tap = (tz * 4 + ty) * 4 + tx;
                    acc = __builtin_fmaf(in_l[sp], s_f[tap], acc);
#pragma unroll
                    for (int s = 1; s < 8; ++s) acc = __builtin_fmaf(in_h[(long long)(s - 1) * dvol + sp], s_f[s * 64 + tap], acc);
                }
            }
        }
        if (a.channel_last) a.out[(((long long)oz * a.t1 + oy) * a.t2 + ox) * a.cs + c] = acc;
        else a.out[idx] = acc;
    }
}

// zero the padding channels [C, cs) of a channel-last grid
__global__ void zero_pad_channels_kernel(float* out, long long nvox, int C, int cs) {
    const int padc = cs - C;
    const long long total = nvox * padc;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
        out[(idx / padc) * cs + C + (idx % padc)] = 0.0f;
}

// Analysis-form kernel shared by the IDWT adjoint and the forward DWT:
//   band_s[c][i] = sum_t src_full[c][2 i + t] * F_s[t],  src_full[u] = src[u - lo] (0 outside [0, n))
struct AnalysisArgs {
    const float* src;      // channel-first (C, n0,n1,n2) or channel-last (n0,n1,n2, cs)
    const float* filt;
    float* band0;          // band 0 of channel c at band0 + c * cstride0
    float* bandh;          // band s>=1 of channel c at bandh + c * cstrideh + (s-1) * dvol
    long long cstride0, cstrideh;
    int C, n0, n1, n2, lo0, lo1, lo2, d0, d1, d2;
    int channel_last, cs;
};

__global__ __launch_bounds__(256) void analysis_kernel(const AnalysisArgs a) {
    __shared__ float s_f[512];
    for (int i = threadIdx.x; i < 512; i += 256) s_f[i] = a.filt[i];
    __syncthreads();
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    const long long total = (long long)a.C * dvol;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ix = (int)(idx % a.d2);
        long long r = idx / a.d2;
        const int iy = (int)(r % a.d1); r /= a.d1;
        const int iz = (int)(r % a.d0);
        const int c = (int)(r / a.d0);
        float acc[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = 0.0f;
#pragma unroll
        for (int tz = 0; tz < 4; ++tz) {
            const int uz = 2 * iz + tz - a.lo0;
            if (uz < 0 || uz >= a.n0) continue;
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                const int uy = 2 * iy + ty - a.lo1;
                if (uy < 0 || uy >= a.n1) continue;
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) {
                    const int ux = 2 * ix + tx - a.lo2;
                    if (ux < 0 || ux >= a.n2) continue;
                    const long long sp = ((long long)uz * a.n1 + uy) * a.n2 + ux;
                    const float v = a.channel_last ? a.src[sp * a.cs + c]
                                                   : a.src[(long long)c * a.n0 * a.n1 * a.n2 + sp];
                    const int tap = (tz * 4 + ty) * 4 + tx;
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc[s] = __builtin_fmaf(v, s_f[s * 64 + tap], acc[s]);
                }
            }
        }
        const long long sp_out = ((long long)iz * a.d1 + iy) * a.d2 + ix;
        a.band0[(long long)c * a.cstride0 + sp_out] = acc[0];
#pragma unroll
        for (int s = 1; s < 8; ++s) a.bandh[(long long)c * a.cstrideh + (long long)(s - 1) * dvol + sp_out] = acc[s];
    }
}

inline int grid_for(long long total, int block = 256, int cap = 256 * 16) {
    long long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

inline int check_level(const void* a, const void* b, const void* c, const void* d, int C, int d0, int d1, int d2,
                       int t0, int t1, int t2, int channel_last, int cs) {
    if (!a || !b || !c || !d) return LFGC_E_NULL;
    if (C < 1 || d0 < 1 || d1 < 1 || d2 < 1 || t0 < 1 || t1 < 1 || t2 < 1) return LFGC_E_SHAPE;
    if (t0 > 2 * d0 + 2 || t1 > 2 * d1 + 2 || t2 > 2 * d2 + 2) return LFGC_E_SHAPE;
    if (channel_last && cs < C) return LFGC_E_SHAPE;
    return LFGC_OK;
}

}  // namespace

extern "C" int lfgc_idwt_level_f32(const float* lll, const float* hf, const float* filter_rev, float* out,
                                   int C, int d0, int d1, int d2, int t0, int t1, int t2,
                                   int channel_last_out, int out_channel_stride, lfgc_stream_t stream) {
    const int rc = check_level(lll, hf, filter_rev, out, C, d0, d1, d2, t0, t1, t2, channel_last_out, out_channel_stride);
    if (rc != LFGC_OK) return rc;
    IdwtArgs a;
    a.lll = lll; a.hf = hf; a.filt = filter_rev; a.out = out;
    a.C = C; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
    a.o0 = (2 * d0 + 2 - t0) / 2; a.o1 = (2 * d1 + 2 - t1) / 2; a.o2 = (2 * d2 + 2 - t2) / 2;
    a.channel_last = channel_last_out; a.cs = out_channel_stride;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)C * t0 * t1 * t2;
    hipLaunchKernelGGL(idwt_level_kernel, dim3(grid_for(total)), dim3(256), 0, st, a);
    LFGC_HIP_CHECK_LAUNCH();
    if (channel_last_out && out_channel_stride > C) {
        const long long nvox = (long long)t0 * t1 * t2;
        hipLaunchKernelGGL(zero_pad_channels_kernel, dim3(grid_for(nvox * (out_channel_stride - C))), dim3(256), 0, st,
                           out, nvox, C, out_channel_stride);
        LFGC_HIP_CHECK_LAUNCH();
    }
    return LFGC_OK;
}

extern "C" int lfgc_idwt_level_bwd_f32(const float* d_out, const float* filter_rev, float* d_lll, float* d_hf,
                                       int C, int d0, int d1, int d2, int t0, int t1, int t2,
                                       int channel_last_out, int out_channel_stride, lfgc_stream_t stream) {
    const int rc = check_level(d_out, filter_rev, d_lll, d_hf, C, d0, d1, d2, t0, t1, t2, channel_last_out, out_channel_stride);
    if (rc != LFGC_OK) return rc;
    AnalysisArgs a;
    a.src = d_out; a.filt = filter_rev; a.band0 = d_lll; a.bandh = d_hf;
    const long long dvol = (long long)d0 * d1 * d2;
    a.cstride0 = dvol; a.cstrideh = 7 * dvol;
    a.C = C; a.n0 = t0; a.n1 = t1; a.n2 = t2;
    a.lo0 = (2 * d0 + 2 - t0) / 2; a.lo1 = (2 * d1 + 2 - t1) / 2; a.lo2 = (2 * d2 + 2 - t2) / 2;
    a.d0 = d0; a.d1 = d1; a.d2 = d2;
    a.channel_last = channel_last_out; a.cs = out_channel_stride;
    hipLaunchKernelGGL(analysis_kernel, dim3(grid_for((long long)C * dvol)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int lfgc_dwt_level_f32(const float* in, const float* filter_fwd, float* out,
                                  int C, int n0, int n1, int n2, lfgc_stream_t stream) {
    if (!in || !filter_fwd || !out) return LFGC_E_NULL;
    if (C < 1 || n0 < 1 || n1 < 1 || n2 < 1) return LFGC_E_SHAPE;
    // _get_padding_size (Torch_Wavelet_Transform.py:59-63): F.pad slots are (last axis lo, hi, ..., first axis
    // lo, hi) while is_odd is indexed first axis first -> the odd bit of axis a pads axis 2-a.
    const int hi0 = 2 + (n2 & 1), hi1 = 2 + (n1 & 1), hi2 = 2 + (n0 & 1);
    AnalysisArgs a;
    a.src = in; a.filt = filter_fwd;
    a.d0 = (n0 + 2 + hi0 - 4) / 2 + 1; a.d1 = (n1 + 2 + hi1 - 4) / 2 + 1; a.d2 = (n2 + 2 + hi2 - 4) / 2 + 1;
    const long long dvol = (long long)a.d0 * a.d1 * a.d2;
    a.band0 = out; a.bandh = out + dvol;
    a.cstride0 = 8 * dvol; a.cstrideh = 8 * dvol;
    a.C = C; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.lo0 = 2; a.lo1 = 2; a.lo2 = 2;
    a.channel_last = 0; a.cs = 0;
    hipLaunchKernelGGL(analysis_kernel, dim3(grid_for((long long)C * dvol)), dim3(256), 0, (hipStream_t)stream, a);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}
