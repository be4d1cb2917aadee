// Do independent f32 VALU instructions issue in the shadow of v_mfma_f32_32x32x16_f16 on gfx950?
// Re-measurement of mfma_interleave.hip with (i) the instruction stream pinned by asm volatile (no compiler
// re-scheduling), (ii) in-kernel cycle stamps (s_memtime) next to the wall clock, because an MFMA-dense loop
// runs well below the nominal clock and wall time alone confuses "more cycles" with "lower clock",
// (iii) random operands, (iv) one and two waves per SIMD.
//   stream per gap:  MFMA ; K x v_fma_f32 (16 rotating independent registers)       [mode 0]
//                    K x v_fma_f32 only                                              [mode 1]
// Output: cycles per gap (median over waves), effective clock, wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define FILL(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2))

template <int K, int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* stamps, const _Float16* src, int iters) {
    extern __shared__ float pad[];
    f32x16 a0 = {0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = src[(threadIdx.x * 8 + i) & 4095]; y[i] = src[(threadIdx.x * 8 + i + 2048) & 4095]; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)x[i & 7] + i;
    float c1 = 0.999f + 1e-6f * threadIdx.x, c2 = 0.001f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < K; ++q) FILL(v[(u * K + q) & 15]);
        }
    }
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + v[r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * NT + threadIdx.x] = s + pad[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int K, int MODE, int NT> void run(float* out, unsigned long long* stamps, const _Float16* src) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 3000, nblk = 256, lds = 100 * 1024;        // 100 KB of LDS: one workgroup per CU
    auto kern = k<K, MODE, NT>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 3; ++w) kern<<<nblk, NT, lds>>>(out, stamps, src, iters);   // warm the clock state
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<nblk, NT, lds>>>(out, stamps, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int nw = nblk * NT / 64;
    std::vector<unsigned long long> h(2 * nw);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10e-9) * 1e-9; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double gaps = (double)iters * 16;
    printf("%s waves/SIMD=%d K=%2d : %7.2f cyc/gap (wave median; x waves/SIMD = %7.2f SIMD cyc/gap-round)  clock %.2f GHz  wall %.3f ms\n",
           MODE == 0 ? "MFMA+VALU" : "VALU only", NT / 256, K, cyc[nw / 2] / gaps, cyc[nw / 2] / gaps, clk[nw / 2], ms);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* stamps; (void)hipMalloc(&stamps, 256 * 8 * 2 * 8);
    std::vector<_Float16> hs(4096);
    srand(7);
    for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.0625f);
    _Float16* src; (void)hipMalloc(&src, 4096 * 2);
    (void)hipMemcpy(src, hs.data(), 4096 * 2, hipMemcpyHostToDevice);
#define ROW(K) run<K, 0, 256>(out, stamps, src); run<K, 0, 512>(out, stamps, src);
    ROW(0) ROW(2) ROW(4) ROW(5) ROW(6) ROW(8) ROW(10) ROW(12) ROW(16) ROW(24)
#define VROW(K) run<K, 1, 256>(out, stamps, src); run<K, 1, 512>(out, stamps, src);
    VROW(4) VROW(8) VROW(16)
    return 0;
}
