/* lfgc.h -- C-ABI of liblfgc.so: the MI355X (gfx950) implementation of the latent-feature-grid
 * sample/decode hot path of Bussler/Latent_Feature_Grid_Compression.
 *
 * The reference has no FFI: its "interface" for this path is the Python nn.Module
 * model/Feature_Grid_Model.py (forward :50-80, decode_volume :102-108, encode_volume :83-99), the
 * wavelet filter wavelet_transform/Torch_Wavelet_Transform.py (encode :75-89, decode :91-104) and
 * the ground-truth sampler data/Interpolation.py:8-44.  Each entry point below names the reference
 * lines it replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer marked "device" is a HIP device pointer owned by the caller (e.g. a torch tensor's
 *     data_ptr()); "host" pointers are ordinary host memory read before the call returns;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream); the
 *     call returns without synchronising; no entry point allocates device memory or synchronises;
 *   - return value: 0 = LFGC_OK, negative = LFGC_E_* argument/shape error (nothing was launched),
 *     positive = the hipError_t of a failed launch.  Never aborts, never throws;
 *   - all floating-point data is IEEE fp32, tensors are dense/contiguous in the stated layout.
 */
#ifndef LFGC_H
#define LFGC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFGC_VERSION 100          /* 0.1.0 */
#define LFGC_MAX_LAYERS 8         /* hidden layers (reference default 4, model/Feature_Grid_Model.py:18) */

enum {
    LFGC_OK = 0,
    LFGC_E_NULL = -1,             /* required pointer is NULL */
    LFGC_E_SHAPE = -2,            /* negative / zero / inconsistent extent */
    LFGC_E_UNSUPPORTED = -3,      /* shape outside the compiled kernel set (see lfgc_mlp_supported) */
    LFGC_E_ALIGN = -4,            /* pointer not 16-byte aligned where required */
    LFGC_E_WORKSPACE = -5         /* workspace / packed buffer too small */
};

typedef void* lfgc_stream_t;      /* hipStream_t */

int lfgc_version(void);
const char* lfgc_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Wavelet transform of the feature grid
 * ---------------------------------------------------------------------------------------------- */

/* One inverse-DWT level.  Replaces _WaveletFilterNd.decode + _unpad_for_reverse
 * (wavelet_transform/Torch_Wavelet_Transform.py:91-104, :69-73) as called per level from
 * Feature_Grid_Model.decode_volume (model/Feature_Grid_Model.py:104-107): torch.cat of the LLL band
 * with the 7 detail bands, grouped conv_transpose3d (stride 2, 4^3 taps), crop to `t`.
 *   lll        device (C, d0,d1,d2)           low band (coarse parameter or previous level's output)
 *   hf         device (C, 7, d0,d1,d2)        detail bands, sub-band s = 4a+2b+c stored at hf[:, s-1]
 *   filter_rev device (8, 4,4,4)              the module's `filter.filter_rev` buffer (fp32)
 *   taps       host float[8] or NULL          the 1-D bank [low taps | high taps] (pywt rec_lo, rec_hi cast to fp32) IF
 *                                             filter_rev is its outer product a[sz][tz]*(a[sy][ty]*a[sx][tx]), the only way the
 *                                             reference builds it (Torch_Wavelet_Transform.py:39-57): the stencil is then
 *                                             contracted axis by axis (224 instead of 512 FMAs per 8 outputs; results equal
 *                                             the dense form up to fp32 rounding) and filter_rev may be NULL
 *   out        device (C, t0,t1,t2)
 * Requires 2*d_a + 2 >= t_a >= 1.  The same `taps` convention applies to every wavelet entry point below. */
int lfgc_idwt_level_f32(const float* lll, const float* hf, const float* filter_rev, const float* taps, float* out,
                        int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream);

/* Adjoint of lfgc_idwt_level_f32 (what autograd derives for the ops above; triggered at
 * training/training.py:137): d_lll (C,d0,d1,d2) and d_hf (C,7,d0,d1,d2) are OVERWRITTEN with the
 * gradients given d_out (C, t0,t1,t2). */
int lfgc_idwt_level_bwd_f32(const float* d_out, const float* filter_rev, const float* taps, float* d_lll, float* d_hf,
                            int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream);

/* Layout conversion of a dense grid between the reference's channel-first (C, V) = decode_volume()'s output
 * (model/Feature_Grid_Model.py:108) and the channel-last (V, channel_stride) form the sampler and the gradient
 * scatter use (V = D*H*W voxels, channel_stride = lfgc_grid_channel_stride(C), pad channels written as 0).
 * to_channel_last != 0: src (C,V) -> dst (V,cs); else src (V,cs) -> dst (C,V).  src != dst. */
int lfgc_grid_layout_f32(const float* src, float* dst, int C, int64_t voxels, int channel_stride,
                         int to_channel_last, lfgc_stream_t stream);

/* The LAST level of decode_volume() written straight in the sampler's channel-last layout, and its adjoint reading the
 * gradient of that layout: lfgc_idwt_level_f32 + lfgc_grid_layout_f32 (and lfgc_grid_layout_f32 +
 * lfgc_idwt_level_bwd_f32) in one pass over the data each (wavelet_transform/Torch_Wavelet_Transform.py:91-104, crop
 * :69-73; the permute is this library's, the reference samples the channel-first grid).
 *   out_cl / d_out_cl  device (t0,t1,t2, channel_stride), channel_stride = lfgc_grid_channel_stride(C); pad channels
 *                      are written as 0 / ignored
 *   taps               REQUIRED (separable bank, see above).  Returns LFGC_E_UNSUPPORTED for taps == NULL, C > 32 or
 *                      arrays of 2^30 bytes and more: the caller then composes the two channel-first entry points. */
int lfgc_idwt_level_cl_f32(const float* lll, const float* hf, const float* taps, float* out_cl,
                           int C, int channel_stride, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream);
int lfgc_idwt_level_cl_bwd_f32(const float* d_out_cl, const float* taps, float* d_lll, float* d_hf,
                               int C, int channel_stride, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream);

/* One forward-DWT level (init only).  Replaces _WaveletFilterNd.encode incl. _pad_for_forward
 * (wavelet_transform/Torch_Wavelet_Transform.py:59-67, :75-89): zero-pad (2, 2 + odd) per axis,
 * grouped conv3d stride 2.  in (C, n0,n1,n2) -> out (C, 8, d0,d1,d2), d_a = (n_a + pad_hi_a) / 2 + 1 - ...
 * exactly: d_a = (n_a + 2 + 2 + odd_a' - 4) / 2 + 1 where odd_a' is the reference's pad-slot quirk
 * (the odd bit of axis a lands on axis 2-a).  filter_fwd device (8,4,4,4); taps = its 1-D bank (pywt dec_lo, dec_hi,
 * each flipped) or NULL. */
int lfgc_dwt_level_f32(const float* in, const float* filter_fwd, const float* taps, float* out,
                       int C, int n0, int n1, int n2, lfgc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pruning ("drop") layers on the wavelet coefficients
 *
 * Every drop layer of the reference multiplies a coefficient tensor (C, ...) by a per-coefficient factor m of
 * shape (...) that is shared by all channels (`x.mul(self.betas.unsqueeze(0))` and relatives):
 *   SmallifyDropout.forward                        model/Smallify_Dropout.py:54-61         m = betas | d_mask
 *   Straight_Through_Dropout.forward               model/Straight_Through_Dropout.py:26-30  m = (rand < mask_values)
 *   MaskedWavelet_Straight_Through_Dropout.forward model/Straight_Through_Dropout.py:54-62  m = sigmoid(mask_values), thr
 *   VariationalDropout.forward                     model/Variational_Dropout_Layer.py:101-112  m = theta + sigma*xi | d_mask
 * `threshold` selects the value rule: NaN -> value = x*m; otherwise the masked straight-through rule
 * value = (x*(m >= threshold) - x*m) + x*m (forward value of the hard mask, gradient of the soft one).
 * In both rules the gradients are d_x = g*m and d_m = sum over channels of g*x.
 * ---------------------------------------------------------------------------------------------- */

/* lfgc_idwt_level_f32 with the drop layers of its inputs folded in (model/Feature_Grid_Model.py:103, :105):
 *   mul_lll device (d0,d1,d2) or NULL     factor of the low band (only the coarsest level has one: drop[0])
 *   mul_hf  device (7, d0,d1,d2) or NULL  factor of the detail bands (drop[level]) */
int lfgc_idwt_level_drop_f32(const float* lll, const float* hf, const float* mul_lll, float threshold_lll,
                             const float* mul_hf, float threshold_hf, const float* filter_rev, const float* taps, float* out,
                             int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream);

/* Adjoint of lfgc_idwt_level_drop_f32: d_lll / d_hf are OVERWRITTEN with the gradients of the UN-multiplied inputs;
 * the factor gradients are ADDED (float atomics over the channels) into d_mul_lll (d0,d1,d2) / d_mul_hf (7,d0,d1,d2),
 * which the caller zero-fills beforehand (NULL = not wanted; wanting them needs lll / hf).
 * penalty_grads: NULL, or a HOST array of 4 device pointers to single floats (each may be NULL) = the upstream gradients
 * of penalty terms whose own gradients are folded into this pass instead of costing a pass of their own:
 *   [0] d(loss)/d(sum lll^2)   -> d_lll += 2 g lll        (only where lll is a parameter: the coarsest level)
 *   [1] d(loss)/d(sum hf^2)    -> d_hf  += 2 g hf         (ΣG² term of SmallifyLoss / VariationalDropoutLoss)
 *   [2] d(loss)/d(sum |mul_lll|), [3] d(loss)/d(sum |mul_hf|) -> d_mul += g sign(mul), for a factor that is itself the
 *       penalised parameter (SmallifyDropout.betas, model/Smallify_Dropout.py:63-64). */
int lfgc_idwt_level_drop_bwd_f32(const float* d_out, const float* filter_rev, const float* taps, const float* lll, const float* hf,
                                 const float* mul_lll, const float* mul_hf, float* d_lll, float* d_hf,
                                 float* d_mul_lll, float* d_mul_hf, const float* const* penalty_grads,
                                 int C, int d0, int d1, int d2, int t0, int t1, int t2, lfgc_stream_t stream);

/* One drop layer on one tensor outside the decode (the layers' own forward(x)): x, out (C, n), mul (n). */
int lfgc_drop_apply_f32(const float* x, const float* mul, float threshold, float* out, int C, int64_t n,
                        lfgc_stream_t stream);
int lfgc_drop_apply_bwd_f32(const float* d_out, const float* x, const float* mul, float* d_x, float* d_mul /* or NULL */,
                            int C, int64_t n, lfgc_stream_t stream);

/* SmallifySignVarianceTracker.sign_variance_pruning_onlyVar (model/Smallify_Dropout.py:106-112) on DEVICE state:
 * phi = sign(betas) - ema; ema += momentum*phi; emavar = (1 - momentum)*(emavar + momentum*phi^2), fp32, the
 * reference's operation order (bit-identical state; the reference moves betas to the CPU every step). */
int lfgc_sign_variance_update_f32(const float* betas, float* ema, float* emavar, float momentum, int64_t n,
                                  lfgc_stream_t stream);
/* The same step for all drop layers of a model in one launch: host arrays of n_layers (<= LFGC_PENALTY_MAX_TERMS)
 * device pointers and lengths. */
int lfgc_sign_variance_update_multi_f32(const float* const* betas, float* const* ema, float* const* emavar,
                                        const int64_t* n, int n_layers, float momentum, lfgc_stream_t stream);

/* Penalty terms of the pruning losses as ONE multi-tensor reduction (SmallifyLoss, model/Smallify_Dropout.py:21-40;
 * VariationalDropoutLoss._collect_penalties + calculate_Dkl, model/Variational_Dropout_Layer.py:48-53, :115-122). */
enum {
    LFGC_PENALTY_L1 = 0,          /* sum |a|                 (l1_loss of betas / mask_values)               */
    LFGC_PENALTY_L2 = 1,          /* sum a^2                 (torch.sum(torch.abs(f) ** 2) of a coefficient tensor) */
    LFGC_PENALTY_DKL = 2          /* sum -k1*sigmoid(k2 + k3*la) + 0.5*softplus(-la) + k1, la = b - 2a (a = log_thetas, b = log_var) */
};
#define LFGC_PENALTY_MAX_TERMS 16
typedef struct lfgc_penalty_term {
    const float* a;               /* device */
    const float* b;               /* device, DKL only */
    int64_t n;
    int32_t kind;
} lfgc_penalty_term;
/* terms: host array; sums: device double[LFGC_PENALTY_SUMS_DOUBLES(n_terms)]: the first n_terms entries receive the
 * results (fp64 accumulation, fixed summation order: bitwise repeatable), the rest is scratch (per-workgroup partials). */
#define LFGC_PENALTY_BLOCKS 1024
#define LFGC_PENALTY_SUMS_DOUBLES(n_terms) ((n_terms) * (1 + LFGC_PENALTY_BLOCKS))
int lfgc_penalty_sums_f32(const lfgc_penalty_term* terms, int n_terms, double* sums, lfgc_stream_t stream);
/* Gradients: grad_a[t] (and grad_b[t] for DKL terms) are OVERWRITTEN with d_sums[t] * d(term t)/d(a | b);
 * d_sums device float[n_terms]; grad_a / grad_b host arrays of device pointers. */
int lfgc_penalty_grads_f32(const lfgc_penalty_term* terms, int n_terms, const float* d_sums,
                           float* const* grad_a, float* const* grad_b, lfgc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Binary checkpoint codec, device side.  The reference's store_model_parameters /
 * restore_model (model/model_utils.py:120-332) do the per-coefficient work in Python lists and strings; these entry
 * points are that work on device buffers.  File layout, header and the hidden-layer / final-layer fields are host code
 * (latent_feature_grid_compression_amd/model/model_utils.py); byte order of every packed stream: MSB first.
 * ---------------------------------------------------------------------------------------------- */

/* Bit mask "1 = coefficient != 0" of x[0..n) (model_utils.py:204-208 + binary_writing :89-107): mask has (n+7)/8 bytes,
 * bit i at byte i/8, MSB first, tail bits zero. */
int lfgc_codec_mask_f32(const float* x, int64_t n, uint8_t* mask, lfgc_stream_t stream);

/* Workspace of the two order-preserving passes below. */
int64_t lfgc_codec_select_workspace_bytes(int64_t n);

/* out[0..*count) = the non-zero values of x in order (model_utils.py:210-212: nonzero + index); count: device int64. */
int lfgc_codec_compact_f32(const float* x, int64_t n, float* out, int64_t* count, void* workspace,
                           int64_t workspace_bytes, lfgc_stream_t stream);

/* out[i] = mask bit (bit_offset + i) ? values[rank of that bit among the set bits of [bit_offset, bit_offset + i)] : 0
 * (restore_model's zero re-insertion, model_utils.py:297-306, there one np.insert per pruned element). */
int lfgc_codec_expand_f32(const uint8_t* mask, int64_t bit_offset, int64_t n, const float* values, float* out,
                          void* workspace, int64_t workspace_bytes, lfgc_stream_t stream);

/* k-entry codebook (k <= 256) of the values x[0..n) by `iterations` Lloyd steps from the SORTED initial centres passed in
 * `centres` (overwritten, stays sorted); labels (uint8, optional) = index of the nearest final centre.  Replaces
 * kmeans_quantization (model_utils.py:65-70: scikit-learn KMeans(n_clusters, n_init=4), unseeded -> the reference's own
 * codebooks are not reproducible; any codebook is a valid file).  Deterministic: per-workgroup partial sums folded in a
 * fixed order. */
/* HOST function (no device work): k sorted initial centres from a SORTED host sample of the values (n <= 2^24) by
 * agglomerative merging of adjacent clusters (Ward's criterion) until k remain; n <= k: the values themselves, padded. */
int lfgc_codec_ward_init_host(const float* sorted_values, int64_t n, int k, float* centres);
#define LFGC_CODEC_KMEANS_PARTS 1024
int64_t lfgc_codec_kmeans_workspace_bytes(int k);
int lfgc_codec_kmeans1d_f32(const float* x, int64_t n, int k, float* centres, uint8_t* labels, int iterations,
                            void* workspace, int64_t workspace_bytes, lfgc_stream_t stream);

/* out[i] = centres[label_i], label_i = bits [bits*i, bits*(i+1)) of `packed`, MSB first, 1 <= bits <= 16
 * (read_in_data_quantized, model_utils.py:255-275). */
int lfgc_codec_dequant_f32(const uint8_t* packed, int64_t packed_bytes, int bits, int64_t n, const float* centres,
                           float* out, lfgc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused sample + Fourier-embed + MLP decoder
 * ---------------------------------------------------------------------------------------------- */

/* Shape of the decoder network (model/Feature_Grid_Model.py:37-48). */
typedef struct lfgc_mlp_desc {
    int32_t grid_channels;   /* C  = feature_grid.shape[0]                                   */
    int32_t hidden;          /* H  = hidden_channel                                           */
    int32_t num_layers;      /* L  = num_layer  (L hidden Linear+SnakeAlt, then final Linear) */
    int32_t n_freqs;         /* FourierEmbedding n_freqs (model/Feature_Embedding.py:20-34)   */
    int32_t d_in;            /* must be 3 */
    int32_t d_out;           /* must be 1 */
} lfgc_mlp_desc;

/* 1 if the compiled kernel set covers this shape: 1 <= C <= 32, 1 <= H <= 128, 1 <= L <= 8, n_freqs == 2, d_in 3, d_out 1
 * (every reference configuration and the NAS ranges of Multi_Objective_NAS.py:127-143); anything else is refused with
 * LFGC_E_UNSUPPORTED by the entry points below. */
int lfgc_mlp_supported(const lfgc_mlp_desc* desc);

/* Channel stride (floats) the sampler expects for the channel-last dense grid: C rounded up to 8. */
int lfgc_grid_channel_stride(int C);

/* Bytes of the packed-parameter blob and of the per-call stash (N samples) for this network. */
int64_t lfgc_packed_bytes(const lfgc_mlp_desc* desc);
int64_t lfgc_stash_bytes(const lfgc_mlp_desc* desc, int64_t n_samples);

/* Re-lay the nn.Linear parameters (weights[l] (out,in) row-major, biases[l] (out), l = 0..L with l = L
 * the final layer; host arrays of L+1 device pointers; named_parameters() order of
 * model/Feature_Grid_Model.py:43-48) into the blob the kernels read.  Run again whenever a parameter
 * changes.  `packed` device, >= lfgc_packed_bytes(), 16-byte aligned. */
int lfgc_pack_mlp_f32(const lfgc_mlp_desc* desc, const float* const* weights, const float* const* biases,
                      float* packed, lfgc_stream_t stream);

/* Where the sample positions come from. */
typedef struct lfgc_positions {
    const float* pos;        /* device (N,3) normalised positions, or NULL to generate the lattice below */
    int64_t n;               /* number of samples when pos != NULL */
    /* pos == NULL: full-volume lattice of visualization/OutputToVTK.py:11-37 (field_from_net), x-slab
     * [x_begin, x_end) of a (res0,res1,res2) volume cut into tiles of `tile` voxels; sample order =
     * row-major (x,y,z) of the slab, i.e. out[(x-x_begin)*res1*res2 + y*res2 + z].  Positions are formed
     * per tile exactly as the reference does (linspace(start,end,n) -> *2-1 -> *scales, fp32) from the ABSOLUTE voxel
     * index, so a slab may start and end anywhere, not only on tile boundaries. */
    int32_t res[3];
    int32_t x_begin, x_end;
    int32_t tile;            /* reference: 32 */
} lfgc_positions;

/* Arithmetic of the layer GEMMs inside lfgc_forward_f32 (inputs, outputs, accumulation and every other operation
 * are fp32 in both):
 *   LFGC_PRECISION_F32    v_mfma_f32_32x32x2_f32, bitwise an fp32 fmaf chain per output;
 *   LFGC_PRECISION_F16X2  every fp32 operand carried as an f16 pair hi + lo (22-24 significant bits), three
 *                         v_mfma_f32_32x32x16_f16 per product block with fp32 accumulation: same error level as the
 *                         fp32 build against the reference (fp32 summation order dominates), several times the
 *                         throughput.  Pre-activations are carried in turns of pi (weight images divided by pi at pack
 *                         time) and SnakeAlt is evaluated with the hardware cosine.  Valid while |pre-activations| <~ 800
 *                         and |grid features| < 65504; beyond that the affected outputs come out as NaN and *status is
 *                         set -- see lfgc_forward_f32.
 *   LFGC_PRECISION_F16    REDUCED precision (opt-in, never the default; does not meet the 1e-5 parity bound): weights and
 *                         activations of the layer GEMMs rounded to f16 (weights pre-scaled per layer as above), ONE
 *                         v_mfma_f32_32x32x16_f16 per product block, fp32 accumulation, fp32 master parameters, fp32
 *                         everything else.  This package's form of the "bf16 compute" BASELINE config 3 names (the
 *                         reference has no reduced-precision path): f16 rather than bf16 because it runs at the same
 *                         MFMA rate with 3 more mantissa bits and the ranges are known (scaled weights, activations
 *                         < 65504, gradients rescaled per tile).  Error ~1e-3 of the output range. */
#define LFGC_PRECISION_F32 0
#define LFGC_PRECISION_F16X2 1
#define LFGC_PRECISION_F16 2

/* forward().  Replaces model/Feature_Grid_Model.py:62-78 (everything after decode_volume):
 * F.grid_sample(bilinear, align_corners=False, zeros) of the dense grid, Embedder.embed
 * (model/Feature_Embedding.py:14-16), torch.cat, L x (Linear + SnakeAlt), final Linear, optional
 * clamp(-1,1) (eval branch :78).
 *   grid_cl   device (D,H,W,Cs) channel-last dense grid, Cs = lfgc_grid_channel_stride(C)
 *   packed    device blob from lfgc_pack_mlp_f32
 *   out       device (N) fp32  (= the (N,1) result)
 *   stash     device, lfgc_stash_bytes(N) bytes, or NULL.  When given, the layer-0 input and every
 *             pre-activation are saved for lfgc_backward_f32 (private layout).
 *   status    device int32 or NULL (ignored by LFGC_PRECISION_F32).  The reference computes in fp32 and stays finite
 *             for any finite parameters (model/Feature_Grid_Model.py:12-13, :72-75); the f16 builds have a range.
 *             With status != NULL the call clears *status (a one-thread kernel), the f16 kernel sets it to 1 if any sample left that range,
 *             and the call then enqueues the same pass on the exact-fp32 build predicated on *status (its workgroups
 *             return immediately when it is 0): `out` (and `stash`) always hold reference-equivalent results, without
 *             a host synchronisation.  With status == NULL out-of-range samples are returned as NaN. */
int lfgc_forward_f32(const lfgc_mlp_desc* desc, const lfgc_positions* positions,
                     const float* grid_cl, int D, int H, int W,
                     const float* packed, int precision, int clamp, float* out, float* stash, int32_t* status,
                     lfgc_stream_t stream);

/* Backward of lfgc_forward_f32 (what autograd derives for model/Feature_Grid_Model.py:62-75;
 * triggered at training/training.py:137).  positions->pos must be non-NULL.
 *   precision   LFGC_PRECISION_* of the data-gradient chain dH_{l-1} = W_l^T dA_l and of the weight-gradient contraction
 *               dW_l = dA_l^T H_{l-1} (f16 builds: f16-split / single f16 products with fp32 accumulation; a tile whose
 *               inputs leave the f16 range takes the exact chain)
 *   stash       device, written by the forward call with the same inputs
 *   d_out       device (N)
 *   d_grid_cl   device (D,H,W,Cs)  ACCUMULATED into with float atomics (caller zeroes it)
 *   d_weights / d_biases  host arrays of L+1 device pointers, each OVERWRITTEN (same shapes as the
 *               parameters)
 *   d_pos       device (N,3) or NULL
 *   workspace   device scratch of lfgc_backward_workspace_bytes() bytes */
int64_t lfgc_backward_workspace_bytes(const lfgc_mlp_desc* desc, int64_t n_samples);
int lfgc_backward_f32(const lfgc_mlp_desc* desc, const lfgc_positions* positions,
                      const float* grid_cl, int D, int H, int W,
                      const float* packed, int precision, const float* stash, const float* d_out,
                      float* d_grid_cl, float* const* d_weights, float* const* d_biases, float* d_pos,
                      void* workspace, int64_t workspace_bytes, lfgc_stream_t stream);

/* The reduced-precision pair under the names SURVEY section 8(b) gives it (BASELINE config 3, "bf16 train step"; the
 * reference itself has no reduced-precision path): exactly lfgc_forward_f32 / lfgc_backward_f32 with precision =
 * LFGC_PRECISION_F16 -- layer GEMMs as single 16-bit products on the matrix pipe, fp32 accumulation, fp32 inputs, outputs
 * and master parameters.  The 16-bit format is IEEE f16, not bfloat16: the same MFMA rate on gfx950, 3 more mantissa bits.
 * What bfloat16 would have over it is fp32's exponent range; the forward entry therefore takes the same `status` word as
 * lfgc_forward_f32: with status != NULL a pass in which any sample left the f16 range (|pre-activation| >~ 800,
 * |grid feature| >= 65504) is redone on the exact-fp32 build, predicated on the word, in stream order -- `out` and
 * `stash` are then finite and reference-equivalent wherever the reference's are, as a bfloat16 path's would be finite.
 * With status == NULL such samples come out as NaN.  The backward needs no word: every tile of dA is rescaled by a power
 * of two before its conversion, and a weight-gradient tile whose activations leave the f16 range takes the exact MFMA. */
int lfgc_forward_bf16(const lfgc_mlp_desc* desc, const lfgc_positions* positions, const float* grid_cl, int D, int H, int W,
                      const float* packed, int clamp, float* out, float* stash, int32_t* status, lfgc_stream_t stream);
int lfgc_backward_bf16(const lfgc_mlp_desc* desc, const lfgc_positions* positions, const float* grid_cl, int D, int H, int W,
                       const float* packed, const float* stash, const float* d_out, float* d_grid_cl,
                       float* const* d_weights, float* const* d_biases, float* d_pos,
                       void* workspace, int64_t workspace_bytes, lfgc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Ground-truth sampler and volume statistics
 * ---------------------------------------------------------------------------------------------- */

/* Training positions for flat voxel indices drawn on the device (IndexDataset.__getitem__, data/IndexDataset.py:90-96
 * with the index table of :56-57 and normalize_volume :7-8): raw (N,3) = integer lattice coordinates as fp32,
 * norm (N,3) = scales * (2 * ((raw - min_idx) / (max_idx - min_idx)) - 1), the reference's fp32 operations (bit-exact).
 *   flat device int64 (N), values in [0, X*Y*Z); res host int32[3]; min_idx / max_idx / scales host float[3]. */
int lfgc_lattice_positions_f32(const int64_t* flat, int64_t n, const int32_t* res, const float* min_idx,
                               const float* max_idx, const float* scales, float* raw, float* norm, lfgc_stream_t stream);

/* The same positions for indices DRAWN by the kernel itself (uniform with replacement, like the randint-on-device sampler
 * of SURVEY section 8 row f2): flat index i of draw `step` = floor(u * X*Y*Z / 2^64), u = first two words (hi, lo) of
 * Philox4x32-10 with counter (i lo, i hi, step lo, step hi) and key (seed lo, seed hi).
 *   state  device int64[2], zeroed ONCE by the caller: [0] = step, advanced by one per call ON THE DEVICE (a captured
 *          launch therefore draws a fresh batch on every graph replay), [1] = scratch, left at 0
 *   flat_out  device int64 (N) or NULL: the drawn indices
 * One launch instead of torch.randint + lfgc_lattice_positions_f32 (+ the generator bookkeeping of a graph replay). */
int lfgc_lattice_sample_f32(uint64_t seed, int64_t* state, int64_t n, const int32_t* res, const float* min_idx,
                            const float* max_idx, const float* scales, float* raw, float* norm, int64_t* flat_out,
                            lfgc_stream_t stream);

/* trilinear_f_interpolation (data/Interpolation.py:8-44), bit-exact: fp32 lattice coordinates, fp64
 * alpha, fp32 lerps in x, y, z order without contraction.
 *   p device (N,3) raw positions; f device (X,Y,Z); min_bb/max_bb/res host float[3]; out device (N). */
int lfgc_gt_interp_f32(const float* p, const float* f, const float* min_bb, const float* max_bb,
                       const float* res, int64_t n, int X, int Y, int Z, float* out, lfgc_stream_t stream);

/* Ground truth and MSE of a train step in one pass (training/training.py:107-109 + nn.MSELoss, :127, :201):
 * gt as lfgc_gt_interp_f32 (bit-exact; also written to gt_out if non-NULL), loss (device float) = mean (pred - gt)^2 with
 * fp64 accumulation in a fixed order, d_pred (N) = 2 (pred - gt) / N = d loss / d pred.
 * workspace: lfgc_gt_mse_workspace_bytes(N) bytes. */
int64_t lfgc_gt_mse_workspace_bytes(int64_t n);
int lfgc_gt_mse_f32(const float* p, const float* f, const float* min_bb, const float* max_bb, const float* res,
                    int64_t n, int X, int Y, int Z, const float* pred, float* gt_out, float* d_pred, float* loss,
                    void* workspace, int64_t workspace_bytes, lfgc_stream_t stream);

/* Partial sums for calculate_deviation_statistics (visualization/OutputToVTK.py:53-60):
 * acc[0] += sum (gt-pred)^2, acc[1] += sum |gt-pred|, acc[2] = min(acc[2], min gt), acc[3] = max(acc[3], max gt)
 * over n elements, fp64 accumulators on device (caller initialises acc = {0, 0, +inf, -inf}). */
int lfgc_deviation_partial_f32(const float* pred, const float* gt, int64_t n, double* acc, lfgc_stream_t stream);

/* Diagnostics: evaluates the kernels' own sin/cos (Cody-Waite + polynomial, lfgc_common.h) and SnakeAlt
 * on n device floats, fast path with the wave-level wide fallback exactly as the hot loops use them.
 * Lets the tests bound the transcendental error against fp64 without going through a network. */
int lfgc_debug_trig_f32(const float* x, int64_t n, float* sin_out, float* cos_out, float* snake_out,
                        lfgc_stream_t stream);

/* Diagnostics: the hardware v_sin_f32 path (fract(x / 2 pi) -> v_sin), NOT used by any kernel; kept so that the
 * accuracy argument for the polynomial path can be re-measured. */
int lfgc_debug_hwsin_f32(const float* x, int64_t n, float* out, lfgc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LFGC_H */
