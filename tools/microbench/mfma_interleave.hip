// How many plain f32 VALU instructions hide under one MFMA when they are INTERLEAVED in the issue stream?
// F32=1: v_mfma_f32_32x32x2_f32 (64-cycle pipe occupancy);  F32=0: v_mfma_f32_32x32x16_f16 (32 cycles).
// 8 waves per block (2 per SIMD), every wave runs {MFMA; K x v_fma} x 16 per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int F32, int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 1e-3f + i); y[i] = (_Float16)(0.5f + i); }
    float fx = threadIdx.x * 1e-3f, fy = 0.75f;
    float v[4] = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (F32) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a0, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) v[q & 3] = __builtin_fmaf(v[q & 3], 1.0001f, 0.5f);
            if (F32) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fy, fx, a1, 0, 0, 0);
            else a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) v[q & 3] = __builtin_fmaf(v[q & 3], 0.9999f, 0.25f);
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int F32, int K, int WAVES> void run(float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    k<F32, K, WAVES><<<256, WAVES * 64>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<F32, K, WAVES><<<256, WAVES * 64>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)(WAVES / 4) * iters * 16;          // MFMAs per SIMD
    printf("%s waves/SIMD=%d  VALU per MFMA=%2d : %.3f ms = %.1f cycles per MFMA @2.3GHz\n", F32 ? "f32 32x32x2 " : "f16 32x32x16",
           WAVES / 4, K, ms, ms * 1e-3 * 2.3e9 / mf);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<1, 0, 8>(out); run<1, 4, 8>(out); run<1, 8, 8>(out); run<1, 12, 8>(out); run<1, 16, 8>(out); run<1, 24, 8>(out);
    run<0, 0, 8>(out); run<0, 2, 8>(out); run<0, 4, 8>(out); run<0, 6, 8>(out); run<0, 8, 8>(out); run<0, 12, 8>(out);
    run<1, 8, 4>(out); run<0, 4, 4>(out);
    return 0;
}
