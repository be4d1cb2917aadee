// Do f16-input MFMAs (v_mfma_f32_32x32x16_f16) and plain f32 VALU work co-execute on one SIMD?
// 8 waves per block (2 per SIMD).  mode 0: all waves MFMA only; 1: all waves VALU only; 2: all waves MFMA then VALU
// (per iteration 16 MFMAs = 512 pipe cycles, and V FMAs); if the pipes are separate mode 2 ~ max(mode0, mode1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int iters) {
    f32x16 a0 = {0}, a1 = {0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 1e-3f + i); y[i] = (_Float16)(0.5f + i); }
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    for (int i = 0; i < iters; ++i) {
        if (mode != 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
            }
        }
        if (mode != 0) {
#pragma unroll
            for (int u = 0; u < V / 4; ++u) {
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
                v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> void run(float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 3; ++mode) {
        k<V><<<256, 512>>>(out, mode, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<V><<<256, 512>>>(out, mode, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("V=%d FMAs/iter mode %d: %.3f ms  (2 waves/SIMD x %d iters x 16 MFMA x 32 cyc = %.3f ms @2.4GHz)\n", V, mode, ms,
               iters, 2.0 * iters * 16 * 32 / 2.4e6);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<128>(out);
    run<256>(out);
    return 0;
}
