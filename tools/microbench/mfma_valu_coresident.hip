// Do an MFMA-only wave and a VALU-only wave that share one SIMD overlap?
// 8 waves per block, 1 block per CU; wave w sits on SIMD w%4, so waves {w, w+4} share a SIMD.
// role by (w>>2): waves 0-3 = "M" (16 MFMAs / iter), waves 4-7 = "V" (V FMAs / iter).
// mode 0: M waves only; 1: V waves only; 2: M and V together (co-resident on every SIMD);
// mode 3: every wave does M then V (same total work per SIMD as mode 2, in lockstep).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int F16, int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool m_role = (wave >> 2) == 0;
    const bool do_mfma = mode == 0 ? m_role : mode == 2 ? m_role : mode == 3;
    const bool do_valu = mode == 1 ? !m_role : mode == 2 ? !m_role : mode == 3;
    const int reps = mode == 3 ? iters / 2 : iters;     // mode 3: each wave does half the iterations of both
    f32x16 a0 = {0}, a1 = {0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 1e-3f + i); y[i] = (_Float16)(0.5f + i); }
    float fx = threadIdx.x * 1e-3f, fy = 1.0f;
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    for (int i = 0; i < reps; ++i) {
        if (do_mfma) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (F16) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fy, fx, a1, 0, 0, 0);
                }
            }
        }
        if (do_valu) {
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

template <int F16, int V> void run(float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 4; ++mode) {
        k<F16, V><<<256, 512>>>(out, mode, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<F16, V><<<256, 512>>>(out, mode, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s V=%d mode %d: %.3f ms\n", F16 ? "f16" : "f32", V, mode, ms);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<1, 128>(out); run<1, 256>(out); run<1, 512>(out);
    run<0, 256>(out); run<0, 512>(out);
    return 0;
}
