// Do f32-input MFMAs (v_mfma_f32_32x32x2_f32) and plain f32 VALU work co-execute on one SIMD?
// 8 waves per block (2 per SIMD), 1 block per CU.  mode 0: even waves MFMA, odd idle; 1: odd waves VALU, even idle;
// 2: even MFMA + odd VALU; 3: every wave MFMA then VALU (same totals as mode 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0 || mode == 2) ? (wave % 2 == 0) : (mode == 3);
    const bool do_valu = (mode == 1 || mode == 2) ? (wave % 2 == 1) : (mode == 3);
    f32x16 a0 = {0}, a1 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f, z = 0.5f, w = 0.25f;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
    for (int i = 0; i < iters; ++i) {
        if (do_mfma) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(z, w, a1, 0, 0, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {          // 256 independent-ish FMAs = 1024 cycles of VALU issue per wave
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
                v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        k<<<256, 512>>>(out, mode, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<256, 512>>>(out, mode, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: one MFMA wave issues iters*16 MFMAs (64 cyc each); one VALU wave issues iters*256 FMAs
        printf("mode %d: %.3f ms  (MFMA-only ideal %.3f ms @2.4GHz)\n", mode, ms, iters * 16 * 64 / 2.4e6);
    }
    return 0;
}
