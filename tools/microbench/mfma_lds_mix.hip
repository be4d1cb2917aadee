// What does the A-operand traffic cost next to the MFMAs?  The forward kernel's layer loop in miniature: per 16-deep
// k-step three v_mfma_f32_32x32x16_f16 on one accumulator chain, R ds_read_b128 of "weights" (row stride 528 B like the
// kernel's images: conflict-free) for the next k-step, and K independent v_fma_f32 per gap; a scheduling fence per gap.
//   MODE 0: operands read from LDS every k-step (as the kernel does)       MODE 1: operands read once (no LDS traffic)
// 8 waves per workgroup (two per SIMD), one workgroup per CU, random data.  Output: wave cycles per gap, clock, wall.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int K, int MODE, int NT>
__global__ __launch_bounds__(NT, 2) void k(float* out, unsigned long long* stamps, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = 132;                                   // row stride in floats (528 B)
    for (int i = threadIdx.x; i < 128 * S; i += NT) lds[i] = src[i & 16383];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const float* row = lds + j * S + 8 * hh;
    f32x16 acc = {0};
    h8 b[8];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) b[q][i] = (_Float16)src[(threadIdx.x * 8 + i + 97 * q) & 16383];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = src[(threadIdx.x + i) & 16383];
    float c1 = 0.999f, c2 = 0.001f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    h8 whi = *reinterpret_cast<const h8*>(row), wlo = *reinterpret_cast<const h8*>(row + 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                h8 nhi = whi, nlo = wlo;
                const float* nsrc = row + 32 * ((m + (ks == 7)) & 3) * S + 16 * ((ks + 1) & 7);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (u == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, b[ks], acc, 0, 0, 0);
                    if (u == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, b[(ks + 1) & 7], acc, 0, 0, 0);
                    if (u == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, b[ks], acc, 0, 0, 0);
                    if (MODE == 0) {
                        if (u == 0) nhi = *reinterpret_cast<const h8*>(nsrc);
                        if (u == 1) nlo = *reinterpret_cast<const h8*>(nsrc + 4);
                    }
#pragma unroll
                    for (int q = 0; q < K; ++q) v[(u * K + q) & 7] = __builtin_fmaf(v[(u * K + q) & 7], c1, c2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                whi = nhi; wlo = nlo;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (lane == 0) {
        const int w = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int K, int MODE, int NT> void run(float* out, unsigned long long* stamps, const float* src) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400, nblk = 256, lds = 128 * 132 * 4 + 70 * 1024;     // > 80 KB: one workgroup per CU
    auto kern = k<K, MODE, NT>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 3; ++w) kern<<<nblk, NT, lds>>>(out, stamps, src, iters);
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
    const double gaps = (double)iters * 96;
    printf("%s waves/SIMD=%d VALU/gap=%d : %6.2f wave-cyc/gap = %6.2f SIMD-cyc per MFMA   clock %.2f GHz  wall %.3f ms\n",
           MODE == 0 ? "LDS operands " : "no LDS reads ", NT / 256, K, cyc[nw / 2] / gaps, cyc[nw / 2] / gaps / (NT / 256), clk[nw / 2], ms);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* stamps; (void)hipMalloc(&stamps, 256 * 8 * 2 * 8);
    std::vector<float> hs(16384);
    srand(7);
    for (auto& v : hs) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    float* src; (void)hipMalloc(&src, 16384 * 4);
    (void)hipMemcpy(src, hs.data(), 16384 * 4, hipMemcpyHostToDevice);
#define ROW(K) run<K, 0, 512>(out, stamps, src); run<K, 1, 512>(out, stamps, src); run<K, 0, 256>(out, stamps, src); run<K, 1, 256>(out, stamps, src);
    ROW(0) ROW(2) ROW(4) ROW(6)
    return 0;
}
