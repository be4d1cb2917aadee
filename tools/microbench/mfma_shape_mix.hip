// Does the MFMA shape change the clock the chip holds under the forward kernel's instruction mix?
// The guide reports bare v_mfma_f32_16x16x32 loops at 1.12-1.15x the FLOP/s of 32x32x16 loops (equal cycles per FLOP,
// higher sustained clock).  The forward kernel is at its power limit, but it is also short of issue
// slots, and a 16x16x32 MFMA holds the vector issue for 8 of its 16 cycles instead of 8 of 32.  This loop has the hidden
// layers' mix per 16 k-values of one 32-sample x 32-row block: the MFMAs of one product (x3 for the hi/lo split), KV
// independent VALU instructions (v_fma_f32 + one v_cos_f32 per 4), 2 ds_read_b128 of operands for a later step,
// with a scheduling fence per MFMA.  Same FLOPs in both shapes:
//   SHAPE 0: 3 x v_mfma_f32_32x32x16_f16 per step            (32 rows x 32 samples x 16 k)
//   SHAPE 1: 6 x v_mfma_f32_16x16x32_f16 per TWO steps' k     (two 16-sample groups x 16 rows ... arranged so that one
//            operand read feeds two MFMAs, as a real kernel would)
// 8 waves per workgroup (two per SIMD), one workgroup per CU, random data.  Output: wall time, SIMD cycles per
// 32x32x16-equivalent MFMA, in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int KV>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* stamps, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = 132;
    for (int i = threadIdx.x; i < 128 * S; i += 512) lds[i] = src[i & 16383];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* row = lds + (lane & 31) * S + 8 * (lane >> 5);
    h8 b[8];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) b[q][i] = (_Float16)src[(threadIdx.x * 8 + i + 97 * q) & 16383];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = src[(threadIdx.x + i) & 16383];
    float c1 = 0.999f, c2 = 0.001f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    f32x16 acc = {0};
    f32x4 a4[4] = {{0}, {0}, {0}, {0}};
    h8 whi = *reinterpret_cast<const h8*>(row), wlo = *reinterpret_cast<const h8*>(row + 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define FILL(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q) & 7]) : "v"(c1), "v"(c2))
#define FILLC(q) asm volatile("v_cos_f32 %0, %0" : "+v"(v[(q) & 7]))
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const float* nsrc = row + 32 * ((ks >> 3) & 3) * S + 16 * ((ks + 1) & 7);
            h8 nhi = whi, nlo = wlo;
            if (SHAPE == 0) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const h8 aa = u == 0 ? wlo : whi;
                    const h8 bb = b[(ks + u) & 7];
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(aa), "v"(bb));
                    if (u == 0) nhi = *reinterpret_cast<const h8*>(nsrc);
                    if (u == 1) nlo = *reinterpret_cast<const h8*>(nsrc + 4);
#pragma unroll
                    for (int q = 0; q < KV; ++q) { if ((u * KV + q) % 4 == 3) FILLC(u * KV + q); else FILL(u * KV + q); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // the same 16 k-values of a 32-row x 32-sample block as 4 MFMAs of 16x16 ... over 32 k: to keep the FLOPs
                // equal per loop trip, 3 products x 2 MFMAs (16 rows x 16 samples x 32 k each) = 3 x 16384 MACs
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const h8 aa = (u >> 1) == 0 ? wlo : whi;
                    const h8 bb = b[(ks + u) & 7];
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a4[u & 3]) : "v"(aa), "v"(bb));
                    if (u == 0) nhi = *reinterpret_cast<const h8*>(nsrc);
                    if (u == 2) nlo = *reinterpret_cast<const h8*>(nsrc + 4);
                    // KV VALU per 32x32x16-equivalent = per two of these MFMAs
                    if (u & 1) {
#pragma unroll
                        for (int q = 0; q < KV; ++q) { if (((u >> 1) * KV + q) % 4 == 3) FILLC((u >> 1) * KV + q); else FILL((u >> 1) * KV + q); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            whi = nhi; wlo = nlo;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int r = 0; r < 4; ++r) s += a4[r][0] + a4[r][1] + a4[r][2] + a4[r][3];
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) {
        const int w = blockIdx.x * 8 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int SHAPE, int KV> void run(float* out, unsigned long long* stamps, const float* src) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 300, nblk = 256, lds = 128 * 132 * 4 + 70 * 1024;
    auto kern = k<SHAPE, KV>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 20; ++w) kern<<<nblk, 512, lds>>>(out, stamps, src, iters);      // ~0.1 s: let the clock settle
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) kern<<<nblk, 512, lds>>>(out, stamps, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const int nw = nblk * 8;
    std::vector<unsigned long long> h(2 * nw);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10e-9) * 1e-9; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double eq = (double)iters * 32 * 3;            // 32x32x16-equivalent MFMAs per wave
    printf("%s  VALU per 32x32x16-equivalent %d : wall %.3f ms   %.1f SIMD cycles per equivalent MFMA   clock %.2f GHz   %.0f equivalent MFMAs / us / SIMD-pair-of-waves\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", KV, ms, cyc[nw / 2] / eq / 2, clk[nw / 2], eq / (ms * 1e3));
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* stamps; (void)hipMalloc(&stamps, 256 * 8 * 2 * 8);
    std::vector<float> hs(16384);
    srand(7);
    for (auto& v : hs) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    float* src; (void)hipMalloc(&src, 16384 * 4);
    (void)hipMemcpy(src, hs.data(), 16384 * 4, hipMemcpyHostToDevice);
#define ROW(K) run<0, K>(out, stamps, src); run<1, K>(out, stamps, src);
    ROW(0) ROW(2) ROW(4) ROW(6)
    return 0;
}
