// Can the two waves that share a SIMD overlap one's MFMA phase with the other's VALU phase?
// 8 waves per block (waves w and w+4 share SIMD w%4), 1 block per CU.  Every wave alternates a block of 16 f16 MFMAs
// (512 matrix-pipe cycles) and a block of V independent-ish f32 FMAs.
//   mode 0: lockstep (all waves: M then V)            mode 1: waves 4-7 start with V (half-period stagger)
//   mode 2: stagger + s_setprio 1 on waves 4-7         mode 3: stagger + s_setprio 1 on waves 0-3
//   mode 4: specialised: waves 0-3 only M (2x), waves 4-7 only V (2x)   mode 5: specialised, roles swapped
//   mode 6: M only (waves 0-3, 2x blocks)   mode 7: V only (waves 4-7, 2x blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int V>
__device__ __forceinline__ void vblock(float& v0, float& v1, float& v2, float& v3) {
#pragma unroll
    for (int u = 0; u < V / 4; ++u) {
        v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
        v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
    }
}
__device__ __forceinline__ void mblock(f32x16& a0, f32x16& a1, h8 x, h8 y) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
    }
}

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    f32x16 a0 = {0}, a1 = {0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 1e-3f + i); y[i] = (_Float16)(0.5f + i); }
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    if (mode == 2 && second) __builtin_amdgcn_s_setprio(1);
    if (mode == 3 && !second) __builtin_amdgcn_s_setprio(1);
    if (mode <= 3) {
        if (mode >= 1 && second) vblock<V>(v0, v1, v2, v3);          // stagger: the second half starts half a period later
        for (int i = 0; i < iters; ++i) {
            mblock(a0, a1, x, y);
            asm volatile("" ::: "memory");
            vblock<V>(v0, v1, v2, v3);
            asm volatile("" ::: "memory");
        }
    } else {
        const bool m_role = (mode == 4 || mode == 6) ? !second : (mode == 5 ? second : false);
        const bool v_role = (mode == 4 || mode == 7) ? second : (mode == 5 ? !second : false);
        for (int i = 0; i < iters; ++i) {
            if (m_role) { mblock(a0, a1, x, y); mblock(a0, a1, x, y); }
            if (v_role) { vblock<V>(v0, v1, v2, v3); vblock<V>(v0, v1, v2, v3); }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> void run(float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 8; ++mode) {
        k<V><<<256, 512>>>(out, mode, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<V><<<256, 512>>>(out, mode, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("V=%d mode %d: %.3f ms\n", V, mode, ms);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<128>(out); run<256>(out);
    return 0;
}
