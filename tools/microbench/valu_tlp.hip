// f32 VALU throughput per SIMD vs waves per SIMD (1, 2, 4, 8) and ILP (independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(float* out, int iters) {
    float v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64 / CHAINS; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) v[c] = __builtin_fmaf(v[c], 1.0001f, 0.5f);
    }
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run(float* out, int waves_per_simd) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, threads = 256 * waves_per_simd;     // one block per CU
    const int blocks = threads > 1024 ? 256 * (threads / 1024) : 256;
    const int tpb = threads > 1024 ? 1024 : threads;
    k<CHAINS><<<blocks, tpb>>>(out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<CHAINS><<<blocks, tpb>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * iters * 64;
    printf("chains %2d waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD (@2.4GHz)\n", CHAINS, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 1024 * 4);
    for (int w : {1, 2, 4, 8}) run<4>(out, w);
    for (int w : {1, 2, 4, 8}) run<16>(out, w);
    return 0;
}
