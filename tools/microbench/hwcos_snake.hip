// SnakeAlt(a) = a/2 + sin(a)^2 = a/2 + 1/2 - cos(2a)/2 with the pre-activation carried in TURNS OF PI (t = a/pi):
//     h = (pi/2) t + 1/2 - cos(2 pi t)/2,      cos(2 pi t) = v_cos_f32(t)   (the hardware takes revolutions)
// Question 1: absolute error of that form on gfx950 against fp64, next to the Cody-Waite + polynomial form the
//             kernels use today (lfgc_snake_t) -- per range of |a|.
// Question 2: issue cost of v_cos_f32 next to v_fma_f32 in a VALU-bound stream (1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../latent_feature_grid_compression_amd/csrc/lfgc_common.h"

__global__ void acc_kernel(const float* t_in, float* h_poly, float* h_hw, float* h_hwsin, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = t_in[i];
    const float a = t * 3.14159274101257324f;                 // what today's kernel would see (one rounding of its own)
    h_poly[i] = lfgc_snake_t<false>(a);
    float cz, sz;
    asm volatile("v_cos_f32 %0, %1" : "=v"(cz) : "v"(t));
    asm volatile("v_sin_f32 %0, %1" : "=v"(sz) : "v"(t));
    const float h0 = __builtin_fmaf(t, 1.57079637050628662f, 0.5f);
    h_hw[i] = __builtin_fmaf(cz, -0.5f, h0);
    h_hwsin[i] = sz;                                            // sin(2 pi t): the derivative's 0.5 + sin(2a)
}

template <int KIND, int NT>
__global__ __launch_bounds__(NT) void thr_kernel(float* out, unsigned long long* stamps, int iters) {
    extern __shared__ float pad[];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.01f * threadIdx.x + i;
    float c1 = 0.999f, c2 = 0.001f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 15]) : "v"(c1), "v"(c2));
            if (KIND == 1) asm volatile("v_cos_f32 %0, %0" : "+v"(v[u & 15]));
            if (KIND == 2) { if (u & 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 15]) : "v"(c1), "v"(c2));
                             else asm volatile("v_cos_f32 %0, %0" : "+v"(v[u & 15])); }
            if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[u & 15]) : "v"(c1));
            if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[u & 15]) : "v"(c1));
            if (KIND == 5) asm volatile("v_fract_f32 %0, %0" : "+v"(v[u & 15]));
            if (KIND == 6) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[u & 15]));
            if (KIND == 7) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(v[u & 15]) : "v"(c1), "v"(c2));
            if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[(2 * u) & 15]) : "v"(*(double*)&v[0]), "v"(*(double*)&v[2]));
            if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[u & 15]) : "v"(c1));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * NT + threadIdx.x] = s + pad[threadIdx.x];
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (NT / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NT> void thr(const char* name, float* out, unsigned long long* stamps) {
    const int iters = 2000, nblk = 256, lds = 100 * 1024;
    auto kern = thr_kernel<KIND, NT>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    kern<<<nblk, NT, lds>>>(out, stamps, iters);
    kern<<<nblk, NT, lds>>>(out, stamps, iters);
    (void)hipDeviceSynchronize();
    const int nw = nblk * NT / 64;
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), stamps, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-28s waves/SIMD=%d : %.2f cyc per wave-instruction (wave), %.2f per SIMD\n", name, NT / 256,
           (double)h[nw / 2] / (iters * 64.0), (double)h[nw / 2] / (iters * 64.0) / (NT / 256));
}

int main() {
    const int n = 1 << 22;
    std::vector<float> t(n), hp(n), hh(n), hs(n);
    float *dt, *dp, *dh, *ds;
    (void)hipMalloc(&dt, n * 4); (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&dh, n * 4); (void)hipMalloc(&ds, n * 4);
    const double PI = 3.14159265358979323846;
    for (double amax : {1.5, 4.0, 10.0, 40.0, 200.0, 2000.0, 60000.0}) {
        unsigned long long s = 12345;
        for (int i = 0; i < n; ++i) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            const double u = ((double)(s >> 11) / 9007199254740992.0) * 2.0 - 1.0;
            t[i] = (float)(u * amax / PI);
        }
        (void)hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice);
        acc_kernel<<<n / 256, 256>>>(dt, dp, dh, ds, n);
        (void)hipMemcpy(hp.data(), dp, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hh.data(), dh, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
        double e_poly = 0, e_hw = 0, e_sin = 0, r_poly = 0, r_hw = 0;
        for (int i = 0; i < n; ++i) {
            const double td = (double)t[i];
            const double truth_t = 0.5 * PI * td + std::pow(std::sin(PI * td), 2);          // exact function of the fp32 t
            const double a32 = (double)(float)((float)t[i] * 3.14159274101257324f);
            const double truth_a = 0.5 * a32 + std::pow(std::sin(a32), 2);                  // exact function of the fp32 a
            e_poly = std::max(e_poly, std::fabs((double)hp[i] - truth_a));
            e_hw = std::max(e_hw, std::fabs((double)hh[i] - truth_t));
            e_sin = std::max(e_sin, std::fabs((double)hs[i] - std::sin(2 * PI * td)));
            r_poly += std::pow((double)hp[i] - truth_a, 2); r_hw += std::pow((double)hh[i] - truth_t, 2);
        }
        printf("|a| <= %8.1f : max abs err  poly %.3e (rms %.2e)   hw-cos-in-turns %.3e (rms %.2e)   hw sin(2 pi t) %.3e   [ulp(h) at max = %.2e]\n",
               amax, e_poly, std::sqrt(r_poly / n), e_hw, std::sqrt(r_hw / n), e_sin, amax * 0.5 * 6e-8);
    }
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* stamps; (void)hipMalloc(&stamps, 256 * 8 * 8);
#define T(K, name) thr<K, 256>(name, out, stamps); thr<K, 512>(name, out, stamps);
    T(0, "v_fma_f32") T(1, "v_cos_f32") T(2, "3 fma : 1 cos") T(3, "v_cvt_pk_f16_f32") T(4, "v_fma_mix_f32") T(5, "v_fract_f32")
    T(6, "v_rndne_f32") T(7, "v_max3_f32 abs") T(8, "v_pk_fma_f32") T(9, "v_mul_f32")
    return 0;
}
