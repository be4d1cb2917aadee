// lfgc_codec.hip -- device side of the binary checkpoint codec (gfx950): the per-coefficient work of the reference's
// store_model_parameters / restore_model (model/model_utils.py:120-332), which there is Python string and list code
// (O(n^2) mask concatenation :207-208, np.insert per pruned element :302-305, scikit-learn k-means on the host).
//   lfgc_codec_mask_f32        bit mask of the non-zero coefficients, MSB first (:204-208, binary_writing :89-107)
//   lfgc_codec_compact_f32     order-preserving removal of the zeros (:210-212)
//   lfgc_codec_kmeans1d_f32    2^bits-entry codebook of a 1-D value set by Lloyd iterations + labels (:65-70, :176-186)
//   lfgc_codec_dequant_f32     labels (`bits` wide, MSB first) -> codebook values (read_in_data_quantized :255-275)
//   lfgc_codec_expand_f32      re-insertion of the zeros by the mask (:297-306)
// Byte / index work, bound by HBM: every kernel streams its input once with coalesced rows; prefix sums are
// block-count + single-block scan + rank-in-block from wave ballots.  Integer results are bit-exact by construction.
#include "lfgc_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPerBlock = 2048;                 // elements per workgroup in the two stream-compaction passes

// ---- mask -----------------------------------------------------------------------------------------------------------
// thread = one output byte = 8 consecutive coefficients (two 16-byte loads when aligned)
__global__ __launch_bounds__(kBlock) void mask_kernel(const float* __restrict__ x, long long n, unsigned char* __restrict__ mask,
                                                      long long nbytes) {
    const long long j = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (j >= nbytes) return;
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long i = j * 8 + k;
        const float v = i < n ? x[i] : 0.0f;
        b |= (unsigned)(v != 0.0f) << (7 - k);
    }
    mask[j] = (unsigned char)b;
}

// ---- order-preserving selection: flag(i) true -> out[rank(i)] ---------------------------------------------------------
// element order inside a workgroup's 2048-element chunk: pass j (0..7) x wave w (0..3) x lane
struct SelectSrc {
    const float* x;              // flag = x[i] != 0                         (compaction)
    const unsigned char* mask;   // flag = bit (bit_offset + i), MSB first   (expansion)
    long long bit_offset;
};

__device__ __forceinline__ bool select_flag(const SelectSrc& s, long long i, long long n) {
    if (i >= n) return false;
    if (s.x) return s.x[i] != 0.0f;
    const long long b = s.bit_offset + i;
    return (s.mask[b >> 3] >> (7 - (int)(b & 7))) & 1;
}

__global__ __launch_bounds__(kBlock) void select_count_kernel(const SelectSrc s, long long n, unsigned* __restrict__ block_counts) {
    const long long base = (long long)blockIdx.x * kPerBlock;
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < kPerBlock / kBlock; ++j) c += select_flag(s, base + j * kBlock + threadIdx.x, n);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    __shared__ unsigned sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

// exclusive scan of the block counts in place (one workgroup; 64-bit offsets), total -> *count
__global__ __launch_bounds__(1024) void scan_kernel(unsigned* __restrict__ counts, long long* __restrict__ offsets,
                                                    long long nblocks, long long* __restrict__ total) {
    __shared__ long long s_wave[16];
    __shared__ long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long b0 = 0; b0 < nblocks; b0 += 1024) {
        const long long b = b0 + threadIdx.x;
        const long long v = b < nblocks ? counts[b] : 0;
        long long incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(incl, off);
            if ((threadIdx.x & 63) >= off) incl += t;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        long long wave_off = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += s_wave[w];
        const long long carry = s_carry;
        if (b < nblocks) offsets[b] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// COMPACT: out[rank] = x[i] for flagged i.   !COMPACT (expand): out[i] = flagged ? values[rank] : 0
template <bool COMPACT>
__global__ __launch_bounds__(kBlock) void select_write_kernel(const SelectSrc s, long long n, const long long* __restrict__ offsets,
                                                              const float* __restrict__ values, float* __restrict__ out) {
    const long long base = (long long)blockIdx.x * kPerBlock;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned s_cnt[8][4];
    bool flag[8];
    unsigned long long bal[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        flag[j] = select_flag(s, base + j * kBlock + threadIdx.x, n);
        bal[j] = __ballot(flag[j]);
        if (lane == 0) s_cnt[j][wave] = (unsigned)__popcll(bal[j]);
    }
    __syncthreads();
    long long run = offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w == wave) {
                const long long i = base + j * kBlock + threadIdx.x;
                const long long rank = run + __popcll(bal[j] & ((1ull << lane) - 1ull));
                if (COMPACT) { if (flag[j]) out[rank] = s.x[i]; }
                else if (i < n) out[i] = flag[j] ? values[rank] : 0.0f;
            }
            run += s_cnt[j][w];
        }
    }
}

// ---- 1-D k-means --------------------------------------------------------------------------------------------------------
// centres are kept sorted: a value belongs to the interval between the midpoints of neighbouring centres, found by
// binary search over the k-1 midpoints in LDS; means of intervals stay ordered, so no re-sort is ever needed.
__device__ __forceinline__ int nearest_centre(const float* s_mid, int k, float v) {
    int lo = 0, hi = k - 1;                      // label = number of midpoints < v
    while (lo < hi) {
        const int m = (lo + hi) >> 1;
        if (s_mid[m] < v) lo = m + 1; else hi = m;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void kmeans_accumulate_kernel(const float* __restrict__ x, long long n, int k,
                                                                   const float* __restrict__ centres,
                                                                   double* __restrict__ part_sum, unsigned* __restrict__ part_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_sum = reinterpret_cast<double*>(s_raw);                 // k
    unsigned* s_cnt = reinterpret_cast<unsigned*>(s_sum + k);         // k
    float* s_mid = reinterpret_cast<float*>(s_cnt + k);               // k (k-1 used)
    for (int j = threadIdx.x; j < k; j += kBlock) {
        s_sum[j] = 0.0; s_cnt[j] = 0u;
        s_mid[j] = j + 1 < k ? 0.5f * (centres[j] + centres[j + 1]) : 3.4e38f;
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float v = x[i];
        const int c = nearest_centre(s_mid, k, v);
        atomicAdd(&s_sum[c], (double)v);
        atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += kBlock) {
        part_sum[(long long)blockIdx.x * k + j] = s_sum[j];
        part_cnt[(long long)blockIdx.x * k + j] = s_cnt[j];
    }
}

__global__ __launch_bounds__(kBlock) void kmeans_update_kernel(int k, int nparts, const double* __restrict__ part_sum,
                                                               const unsigned* __restrict__ part_cnt, float* __restrict__ centres) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= k) return;
    double s = 0.0;
    unsigned long long c = 0;
    for (int p = 0; p < nparts; ++p) { s += part_sum[(long long)p * k + j]; c += part_cnt[(long long)p * k + j]; }
    if (c > 0) centres[j] = (float)(s / (double)c);        // an empty cluster keeps its centre
}

__global__ __launch_bounds__(kBlock) void kmeans_label_kernel(const float* __restrict__ x, long long n, int k,
                                                              const float* __restrict__ centres, unsigned char* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float* s_mid = reinterpret_cast<float*>(s_raw);
    for (int j = threadIdx.x; j < k; j += kBlock) s_mid[j] = j + 1 < k ? 0.5f * (centres[j] + centres[j + 1]) : 3.4e38f;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
        labels[i] = (unsigned char)nearest_centre(s_mid, k, x[i]);
}

// ---- dequantisation -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void dequant_kernel(const unsigned char* __restrict__ packed, long long packed_bytes,
                                                         int bits, long long n, const float* __restrict__ centres,
                                                         float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    unsigned label;
    if (bits == 8) {
        label = packed[i];
    } else {                                     // bits [bits*i, bits*(i+1)) of the stream, MSB first (<= 16 bits: 3 bytes)
        const long long b0 = i * bits;
        const long long byte0 = b0 >> 3;
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) w = (w << 8) | (byte0 + k < packed_bytes ? packed[byte0 + k] : 0u);
        label = (w >> (24 - (int)(b0 & 7) - bits)) & ((1u << bits) - 1u);
    }
    out[i] = centres[label];
}

inline unsigned blocks_cap(long long n, int per, long long cap) {
    long long g = (n + per - 1) / per;
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

int run_select(const SelectSrc& s, long long n, const float* values, float* out, long long* count, void* workspace,
               long long workspace_bytes, bool compact, hipStream_t stream) {
    const long long nblocks = (n + kPerBlock - 1) / kPerBlock;
    if (workspace_bytes < lfgc_codec_select_workspace_bytes(n)) return LFGC_E_WORKSPACE;
    if (nblocks > 0x7fffffffLL) return LFGC_E_UNSUPPORTED;
    long long* offsets = reinterpret_cast<long long*>(workspace);
    unsigned* counts = reinterpret_cast<unsigned*>(offsets + nblocks + 1);
    long long* total = count ? count : offsets + nblocks;
    hipLaunchKernelGGL(select_count_kernel, dim3((unsigned)nblocks), dim3(kBlock), 0, stream, s, n, counts);
    LFGC_HIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, counts, offsets, nblocks, total);
    LFGC_HIP_CHECK_LAUNCH();
    if (compact) hipLaunchKernelGGL(select_write_kernel<true>, dim3((unsigned)nblocks), dim3(kBlock), 0, stream, s, n, offsets, values, out);
    else hipLaunchKernelGGL(select_write_kernel<false>, dim3((unsigned)nblocks), dim3(kBlock), 0, stream, s, n, offsets, values, out);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

}  // namespace

extern "C" int lfgc_codec_mask_f32(const float* x, int64_t n, uint8_t* mask, lfgc_stream_t stream) {
    if (!x || !mask) return LFGC_E_NULL;
    if (n < 1) return LFGC_E_SHAPE;
    const long long nbytes = (n + 7) / 8;
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((nbytes + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       x, (long long)n, mask, nbytes);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}

extern "C" int64_t lfgc_codec_select_workspace_bytes(int64_t n) {
    if (n < 1) return 16;
    const long long nblocks = (n + kPerBlock - 1) / kPerBlock;
    return (nblocks + 1) * 8 + nblocks * 4 + 16;
}

extern "C" int lfgc_codec_compact_f32(const float* x, int64_t n, float* out, int64_t* count, void* workspace,
                                      int64_t workspace_bytes, lfgc_stream_t stream) {
    if (!x || !out || !count || !workspace) return LFGC_E_NULL;
    if (n < 1) return LFGC_E_SHAPE;
    SelectSrc s; s.x = x; s.mask = nullptr; s.bit_offset = 0;
    return run_select(s, n, nullptr, out, reinterpret_cast<long long*>(count), workspace, workspace_bytes, true, (hipStream_t)stream);
}

extern "C" int lfgc_codec_expand_f32(const uint8_t* mask, int64_t bit_offset, int64_t n, const float* values, float* out,
                                     void* workspace, int64_t workspace_bytes, lfgc_stream_t stream) {
    if (!mask || !values || !out || !workspace) return LFGC_E_NULL;
    if (n < 1 || bit_offset < 0) return LFGC_E_SHAPE;
    SelectSrc s; s.x = nullptr; s.mask = mask; s.bit_offset = bit_offset;
    return run_select(s, n, values, out, nullptr, workspace, workspace_bytes, false, (hipStream_t)stream);
}

extern "C" int64_t lfgc_codec_kmeans_workspace_bytes(int k) {
    if (k < 1) return 0;
    return (int64_t)LFGC_CODEC_KMEANS_PARTS * k * (8 + 4) + 64;
}

extern "C" int lfgc_codec_kmeans1d_f32(const float* x, int64_t n, int k, float* centres, uint8_t* labels, int iterations,
                                       void* workspace, int64_t workspace_bytes, lfgc_stream_t stream) {
    if (!x || !centres || !workspace) return LFGC_E_NULL;
    if (n < 1 || k < 1 || k > 256 || iterations < 0) return LFGC_E_SHAPE;
    if (workspace_bytes < lfgc_codec_kmeans_workspace_bytes(k)) return LFGC_E_WORKSPACE;
    const unsigned parts = blocks_cap(n, kBlock * 8, LFGC_CODEC_KMEANS_PARTS);
    double* part_sum = reinterpret_cast<double*>(workspace);
    unsigned* part_cnt = reinterpret_cast<unsigned*>(part_sum + (size_t)LFGC_CODEC_KMEANS_PARTS * k);
    const int lds = k * (8 + 4 + 4);
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(parts), dim3(kBlock), lds, (hipStream_t)stream,
                           x, (long long)n, k, centres, part_sum, part_cnt);
        LFGC_HIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(kmeans_update_kernel, dim3((k + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                           k, (int)parts, part_sum, part_cnt, centres);
        LFGC_HIP_CHECK_LAUNCH();
    }
    if (labels) {
        hipLaunchKernelGGL(kmeans_label_kernel, dim3(blocks_cap(n, kBlock * 4, 4096)), dim3(kBlock), k * 4, (hipStream_t)stream,
                           x, (long long)n, k, centres, labels);
        LFGC_HIP_CHECK_LAUNCH();
    }
    return LFGC_OK;
}

extern "C" int lfgc_codec_dequant_f32(const uint8_t* packed, int64_t packed_bytes, int bits, int64_t n, const float* centres,
                                      float* out, lfgc_stream_t stream) {
    if (!packed || !centres || !out) return LFGC_E_NULL;
    if (n < 1 || bits < 1 || bits > 16 || packed_bytes * 8 < n * (int64_t)bits) return LFGC_E_SHAPE;
    hipLaunchKernelGGL(dequant_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       packed, (long long)packed_bytes, bits, (long long)n, centres, out);
    LFGC_HIP_CHECK_LAUNCH();
    return LFGC_OK;
}
