/* isegprobe_hip.h -- C ABI of libisegprobe_hip.so: the MI355X (gfx950) kernels behind the
 * iSegProbe per-click dense-feature path (click maps -> ViT featurizer -> upsampler -> conv
 * seg head -> click-map fusion).
 *
 * Conventions (every entry point):
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*;
 *   - stateless: never allocates, never synchronises, no globals; the caller owns every
 *     buffer (including workspaces) and the stream; safe from any host thread;
 *   - returns ISP_OK (0) or a negative ISP_ERR_* code; nothing is launched on error;
 *   - "bf16" buffers hold raw bfloat16 bits (uint16); activations are NHWC / token-major;
 *   - dense weights use the nn.Linear / flattened Conv2d layout Wt[N][K], bf16.
 *
 * The reference (havrylovv/iSegProbe) has exactly one native entry point,
 *   get_dist_maps(points f32[2P,3], H, W, norm_delimeter) -> f32[2,H,W]
 *   (core/utils/cython/_get_dist_maps.pyx:18-21, called from core/model/ops.py:21-34);
 * isp_click_maps_fwd(..., round_clicks=1) is its drop-in.  Every other function replaces a
 * torch op sequence of the reference's Python path; the file:line each one replaces is
 * cited on its declaration.  INTEGRATION.md shows the ctypes binding for each.
 */
#ifndef ISEGPROBE_HIP_H
#define ISEGPROBE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ISP_OK 0
#define ISP_ERR_INVALID (-1)     /* bad pointer / dimension / alignment */
#define ISP_ERR_UNSUPPORTED (-2) /* valid request this build has no kernel for */
#define ISP_ERR_LAUNCH (-3)      /* HIP refused the launch */

#define ISP_F32 0
#define ISP_BF16 1
#define ISP_F16 2 /* IEEE half */

/* ABI version; bumped on any signature change. */
int isp_abi_version(void);

/* ---- click maps: DistMaps.get_coord_features, core/model/ops.py:35-77 (torch path, exact
 * fp32 op order, fractional clicks kept) and, with round_clicks=1, get_dist_maps
 * (_get_dist_maps.pyx:18-64: clicks rounded half-to-even, row<0 = empty).
 * points [B,2P,3] f32 (row, col, order; first P rows positive), out [B,2,H,W] f32. */
int isp_click_maps_fwd(const float* points, float* out, int B, int P, int H, int W, float norm_radius,
                       float spatial_scale, int use_disks, int round_clicks, void* stream);

/* ---- BatchImageNormalize + prev-mask split: ops.py:96-105, iseg_base_model.py:91-98.
 * image [B,in_ch(3|4),H,W] f32 -> out [B,3,H,W] f32, prev_mask [B,1,H,W] f32 (may be NULL).
 * mean3/std3 are HOST pointers to 3 floats. */
int isp_normalize_fwd(const float* image, float* out, float* prev_mask, int B, int in_ch, int H, int W,
                      const float* mean3, const float* std3, void* stream);

/* ---- patch matrix for the fused image+click patch-embed GEMM: the im2col side of
 * dinov2/layers/patch_embed.py:71-87 and featurizers/utils/patch_embed.py:37-42 joined as in
 * DINOv2.py:518-523.  A [B*h*w, Kpad] bf16, row = [img(n_img,p,p) | prev(n_prev,p,p) |
 * maps(n_maps,p,p) | 0...]. */
int isp_patchify_fwd(const float* image, const float* prev_mask, const float* click_maps, void* A_bf16, int B, int H,
                     int W, int patch, int n_img, int n_prev, int n_maps, int Kpad, void* stream);

/* ---- fused GEMM epilogues */
#define ISP_EP_BIAS_BF16 0      /* out bf16 = v + bias                                   */
#define ISP_EP_BIAS_RELU_BF16 1 /* out bf16 = relu(v + bias)        ConvModule, conv_heads.py:59 */
#define ISP_EP_BIAS_GELU_BF16 2 /* out bf16 = gelu_erf(v + bias)    Mlp, mlp.py:34-40     */
#define ISP_EP_BIAS_F32 3       /* out f32  = v + bias                                   */
#define ISP_EP_RESIDUAL_F32 4   /* out f32 += gamma * (v + bias)    LayerScale + residual, block.py:92-117 */
#define ISP_EP_TOKENS_F32 5     /* out f32 [b*(T+1)+1+t] = v + bias + pos[1+t]   DINOv2.py:523-528 */
#define ISP_EP_AXPY_RES_BF16 6  /* out bf16 = res + alpha * (v + bias)   FeatUp JBUStack fix-up  */
#define ISP_EP_BIAS_TAPS_RELU_BF16 7 /* conv3x3 only: relu(v + bias - sum_{taps outside the image} pos[t][n]);
                                        a per-pixel affine map folded into the conv (see gemm.hip) */

#define ISP_EP_RELU_DOT_PARTIAL_F32 8 /* conv/gemm + 1x1 classifier fused: out f32 partial[slot][M] =
                                         sum_n relu(v + bias[n]) * gamma[n] over the slot's channels;
                                         slots = isp_conv3x3_partial_slots(N); close with isp_sum_partials_f32 */

#define ISP_EP_BIAS_QGELU_BF16 9 /* out bf16 = quick_gelu(v + bias) = x*sigmoid(1.702x)   CLIP MLP, maskclip/model.py:231-233 */
#define ISP_EP_BIAS_GELU_SAVE_BF16 10 /* training forward of Mlp.fc1: out bf16 = gelu_erf(v + bias) and out2 bf16 = v + bias
                                       * (the pre-activation the backward needs; mlp.py:34-40 under autograd) */
#define ISP_EP_BIAS_QGELU_SAVE_BF16 12 /* as 10 with QuickGELU (CLIP ResidualAttentionBlock.mlp, maskclip/model.py:231-233) */
#define ISP_EP_MUL_DQGELU_BF16 13      /* as 11 with QuickGELU' */
#define ISP_EP_MUL_DGELU_BF16 11 /* backward of GELU fused into the fc2 data-gradient GEMM: out bf16 = v * gelu'(res),
                                  * res bf16 = the saved pre-activation (row stride ldo) */

#define ISP_EP_AXPY_RES_STATS_BF16 14 /* isp_gemm_f16, N <= 448: ISP_EP_AXPY_RES_BF16 that also writes, per output row, the sum and the sum of
                                       * squares of the stored values: out2 f32 [isp_gemm_stats_slots()][M][2] (partial per wave column group) */
#define ISP_EP_LNFOLD_BF16 15      /* isp_gemm_f16: out = rstd[m] * (v - mean[m] * gamma[n]) + bias[n]: a LayerNorm over the first
                                    * tokens_per_image (= D) columns of A folded into this GEMM (weights carry the LayerNorm gain, gamma[n]
                                    * = their row sums, bias = c + W b); row statistics from res = the producer's partial sums
                                    * [img_h slots][M][2] (img_h <= 8), alpha = the LayerNorm epsilon (loftup/layers.py:186-228, 53-58) */
#define ISP_EP_LNFOLD_GELU_BF16 16 /* the same followed by GELU (erf form; its sigmoid fit, max |error| 2.5e-5, on half outputs): FeedForward's first layer */

#define ISP_EP_RESIDUAL_STATS_F32 17 /* isp_gemm_f16: ISP_EP_RESIDUAL_F32 that also writes out3 = IEEE-half copy of the updated rows (row stride
                                      * ldo) and out2 = their per-row partial sums f32 [isp_gemm_f16_stats_slots(M, N)][M][2]: the ViT block's
                                      * LayerNorms (block.py:92-117) are then folded into the qkv / fc1 GEMMs (ISP_EP_LNFOLD_*) */

#define ISP_EP_LNFOLD_LAYERNORM_BF16 18 /* isp_gemm_f16, N <= 512: ISP_EP_LNFOLD_BF16 followed by a LayerNorm over the N output
                                      * channels of the row (gain `pos`, bias `out2` as const float*, epsilon `alpha2`), computed in the
                                      * epilogue of a tile that spans the whole row: LoftUp's tail LayerNorm -> 1x1 conv -> channel
                                      * LayerNorm (loftup/loftup.py:139-149) as ONE GEMM */

#define ISP_EP_BIAS_RELU_STATS_BF16 19 /* isp_conv3x3_nhwc_f16: ISP_EP_BIAS_RELU_BF16 that also writes out2 = per-pixel partial sums f32
                                      * [isp_conv_stats_slots(N)][B*H*W][2] of the stored values (for ISP_EP_LNFOLD_* consumers) */

typedef struct isp_epilogue {
    int kind;             /* ISP_EP_* */
    void* out;            /* bf16 or f32 per kind */
    long ldo;             /* row stride of out in elements (0 = N) */
    const float* bias;    /* [N] or NULL */
    const float* gamma;   /* [N] LayerScale or NULL (RESIDUAL) */
    const float* pos;     /* [(T+1), ldo] interpolated pos-embed or NULL (TOKENS); [9, N] tap table (BIAS_TAPS) */
    int tokens_per_image; /* T (TOKENS) */
    const void* res;      /* bf16 [M, ldo] residual (AXPY_RES) */
    float alpha;          /* (AXPY_RES) */
    int img_h, img_w;     /* image extent (BIAS_TAPS) */
    void* out2;           /* second output (BIAS_GELU_SAVE), row stride ldo; row statistics (..._STATS) */
    void* out3;           /* third output: the 16-bit copy of RESIDUAL_STATS */
    float alpha2;         /* second scalar: epsilon of the output LayerNorm (LNFOLD_LAYERNORM) */
} isp_epilogue;

/* ---- C = A . Wt^T with a fused epilogue.  A [M, lda>=K] bf16, Wt [N,K] bf16, K % 64 == 0,
 * N % 4 == 0.  Replaces nn.Linear in attention.py:54-71, mlp.py:34-40, the patch-embed convs
 * (as GEMMs) and every 1x1 conv on the path. */
int isp_gemm_bf16(const void* A, long lda, const void* Wt, long M, int N, int K, const isp_epilogue* ep, void* stream);
/* The same on IEEE-half operands with 16-bit outputs in half (LoftUp's inference stream, loftup/layers.py:160-228): epilogue
 * kinds ISP_EP_BIAS_BF16, ISP_EP_BIAS_GELU_BF16 (half outputs saturate at +-65504), ISP_EP_AXPY_RES_BF16 (res and out half)
 * and ISP_EP_RESIDUAL_F32 (the ViT's fp32 residual stream); others ISP_ERR_UNSUPPORTED. */
int isp_gemm_f16(const void* A, long lda, const void* Wt, long M, int N, int K, const isp_epilogue* ep, void* stream);
int isp_conv_stats_slots(int N); /* partial-statistics slots of ISP_EP_BIAS_RELU_STATS_BF16 for N output channels */
int isp_gemm_stats_slots(void); /* partial-statistics slots of ISP_EP_AXPY_RES_STATS_BF16 */
int isp_gemm_f16_stats_slots(long M, int N); /* ... of ISP_EP_RESIDUAL_STATS_F32 for an M x N problem */

/* ---- 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on NHWC bf16.
 * in [B,H,W,C] (C % 64 == 0), Wt [N][9*C] with K index = ((ky*3+kx)*C + c), i.e.
 * conv.weight.permute(0,2,3,1).  Replaces ConvModule/Conv2d 3x3 in heads/conv_heads.py:51-73,
 * loftup/loftup.py:53-63 and LiFT.py:12-27. */
int isp_conv3x3_nhwc_bf16(const void* in, const void* Wt, int B, int H, int W, int C, int N, const isp_epilogue* ep,
                          void* stream);
/* The same with IEEE-half operands and 16-bit outputs (epilogue kinds ISP_EP_BIAS_BF16, ISP_EP_BIAS_RELU_BF16,
 * ISP_EP_BIAS_TAPS_RELU_BF16 then write half; ISP_EP_RELU_DOT_PARTIAL_F32 as before): ConvSegHead's convolutions
 * (heads/conv_heads.py:51-73) behind the FeatUp-JBU stack, whose maps are half.  N % 192 == 0, ldo % 8 == 0, 16-byte
 * aligned output; anything else returns ISP_ERR_UNSUPPORTED (callers convert to bf16 and use the entry above). */
int isp_conv3x3_nhwc_f16(const void* in, const void* Wt, int B, int H, int W, int C, int N, const isp_epilogue* ep,
                         void* stream);

int isp_conv3x3_partial_slots(int N);

/* ---- First head convolution taken through the bilinear resize: relu(conv3x3(F.interpolate(x, (H, W), bilinear,
 * align_corners=True)) + bias) without the [B,H,W,C] map -- iseg_probe_model.py:120-129 (or the bilinear upsampler plugin,
 * basic_upsamplers.py:28-33) followed by heads/conv_heads.py:59-73.  The convolution is linear in the resized map, so
 *   out[p][n] = act(bias[n] + sum_t [p+t inside] sum_{q in 2x2(p+t)} a(p+t, q) Z[q][t*N + n]),   Z = x [B*h*w, C] x [W_0..W_8]^T
 * with Z ONE low-resolution GEMM (isp_gemm_f16 / isp_gemm_bf16 against Wz[t*N + n][c] = conv.weight[n][c][ty][tx]) and this
 * entry the blend: 36 multiply-adds per output value instead of 9*C.  z [B*h*w, 9*N] IEEE half (N % 64 == 0) or f32 (N % 32 == 0;
 * the fp32-accurate checking mode), bias f32 [N] or NULL, out [B,H,W,N] half / bf16 / f32 (f32 only for f32 z).
 * isp_conv3x3_of_bilinear_supported: 1 when every 16 x 16 output tile's tap neighbourhood touches at most 5 x 5 source pixels
 * (up-scaling by about 5.7 or more); otherwise callers materialise the resized map and use isp_conv3x3_nhwc_*. */
int isp_conv3x3_of_bilinear_supported(int h, int w, int H, int W, int N, int z_dtype);
int isp_conv3x3_of_bilinear_blend(const void* z, int z_dtype, const float* bias, void* out, int out_dtype, int B, int h, int w,
                                  int H, int W, int N, int relu, void* stream);
/* Training: adjoint of that blend w.r.t. the tap planes.  g [B,H,W,N] bf16 = gradient of the PRE-activation (the caller has
 * applied the ReLU mask, e.g. isp_relu_mask_colsum, which also yields the bias gradient); dz [B*h*w, 9*N] bf16 =
 * sum_p [p+t inside] a(p+t, q) g[p][n] -- from which dx = dz Wz and dWz = dz^T x are two small GEMMs at low resolution (autograd of
 * trainer.py:214-232 through iseg_probe_model.py:120-129 + conv_heads.py:59-73).  Gathers, nothing atomic.  N % 8 == 0;
 * workspace: isp_conv3x3_of_bilinear_bwd_workspace_bytes(B, h, W, N) bytes (the row-adjoint, fp32). */
long isp_conv3x3_of_bilinear_bwd_workspace_bytes(int B, int h, int W, int N);
int isp_conv3x3_of_bilinear_blend_bwd(const void* g_bf16, void* dz_bf16, void* workspace, int B, int h, int w, int H, int W, int N,
                                      void* stream);
int isp_sum_partials_f32(const float* partial, float* out, long M, int slots, float bias, void* stream);

/* ---- LayerNorm over the last dim (fp32 statistics), nn.LayerNorm(eps) of DINOv2.py:98 and
 * loftup/layers.py.  group_out>0 drops `skip` leading rows of every (group_out+skip)-row
 * group of the input (cls-token drop, DINOv2.py:533-534).  ld_in / ld_out: row strides in
 * elements (0 = D); output columns [D, ld_out) are zero-filled (channel padding for the GEMMs). */
int isp_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, long rows, int D, float eps,
                      int in_dtype, int out_dtype, int group_out, int skip, long ld_in, long ld_out, void* stream);

/* ---- softmax(Q K^T * scale) V, head_dim 64, 128 or 256, bf16 in/out, fp32 online softmax; never
 * materialises the score matrix.  Element strides are explicit so the packed qkv tensor of
 * attention.py:56-60 is consumed in place.  Q [B,Lq,H,hd], K/V [B,Lk,H,hd], O [B,Lq,H,hd].
 * Also replaces nn.MultiheadAttention in LoftUp's CrossAttentionLayer (loftup/layers.py:182-198),
 * head_dim 101 zero-padded to 128 (n_dim 384) or 197 zero-padded to 256 (n_dim 768, BASELINE configs[4]). */
int isp_attention_fwd(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, int head_dim,
                      long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b, long kv_stride_l,
                      long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h, float scale, void* stream);

/* head_dim 64 only (else ISP_ERR_UNSUPPORTED): Q already carries scale * log2(e), i.e. Q K^T are base-2 logits and the
 * kernel's exponentials take the MFMA output as it is.  The ViT trunk folds the factor into the Q rows of its packed qkv
 * weights and bias (attention.py:62 multiplies q by the scale after the projection: same product, rounded once). */
int isp_attention_fwd_logit2(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                             int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                             long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                             void* stream);

/* IEEE-half Q, K, V, O on the generic kernel (head_dim 64 / 128 / 256): LoftUp's cross-attention in its half-precision
 * inference stream */
int isp_attention_fwd_f16(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, int head_dim,
                          long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b, long kv_stride_l,
                          long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h, float scale, void* stream);

/* Software-pipelined forward (csrc/attention_pipe.hip): one wave per SIMD, 64 queries per wave as two streams that share
 * every K / V fragment read, QK^T of tile t+1 and the PV product of tile t interleaved with the softmax of tile t.
 * head_dim 64 (Attention.forward, dinov2/layers/attention.py:54-71) or 128 (LoftUp's nn.MultiheadAttention,
 * loftup/layers.py:182-198, head_dim 101 zero-padded).  Q carries softmax scale x log2(e) (as isp_attention_fwd_logit2);
 * bf16 (f16 = 0) or IEEE-half operands.  Inference only.  isp_attention_pipe_supported tells whether a problem is taken
 * (Lk >= 128, 32-bit addressable K / V slice, ISEGPROBE_ATT_PIPE != 0); isp_attention_fwd_logit2[_f16] route to it. */
int isp_attention_pipe_supported(int head_dim, int Lq, int Lk, long kv_stride_l);
int isp_attention_fwd_pipe(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, int head_dim,
                           long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b, long kv_stride_l,
                           long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h, int f16, void* stream);

/* isp_attention_fwd_logit2 on IEEE-half Q, K, V, O (the ViT trunk's half-precision inference stream, head_dim 64; LoftUp's, head_dim 128 / 256) */
int isp_attention_fwd_logit2_f16(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                 int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                                 long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                                 void* stream);

/* Training variant: also writes lse[b*H+h][q] (row stride lse_ld >= Lq, fp32) = log2 sum_k exp2(s_qk * scale * log2 e),
 * the statistic isp_attention_bwd needs to recompute the probabilities. */
int isp_attention_fwd_lse(const void* Q, const void* K, const void* V, void* O, float* lse, long lse_ld, int B, int H, int Lq,
                          int Lk, int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                          long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                          float scale, void* stream);

/* ---- Backward of isp_attention_fwd (head_dim 64, 128 or 256): dQ (nullable: skipped), dK, dV (bf16, strides of Q / of K,V) from O, dO (strides
 * o_stride_*) and lse.  delta is a [B*H, stat_ld] fp32 workspace (rowsum(dO*O)); stat_ld % 64 == 0 is the row
 * stride of BOTH lse and delta.  What autograd does for Attention.forward (dinov2/layers/attention.py:54-71) when the
 * reference trains with feats_injection_mode="before_backbone" (models/sbd/dinov2/patch-embed_*.py:40); the
 * probability matrix is recomputed per 64x64 tile, never stored.  kv_split_workspace (nullable): 2*B*Lk*H*head_dim fp32;
 * when given and the key side alone cannot fill the chip (few keys, many queries: LoftUp's cross-attention at the training
 * crop) the query range is split over several blocks that add fp32 partials of dK / dV into it. */
int isp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                      float* delta, long stat_ld, void* dQ, void* dK, void* dV, int B, int H, int Lq, int Lk, int head_dim,
                      long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b, long kv_stride_l,
                      long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h, float scale,
                      float* kv_split_workspace, void* stream);

/* ---- LayerNorm backward w.r.t. the input (frozen affine): gx (fp32, row stride ld_gx) (+)= dLN(x; gamma)(gy),
 * statistics recomputed from x (fp32 or bf16, row stride ld_x, first D columns); gy bf16 (row stride ld_gy); optional
 * bf16 copy of the updated gx (row stride ld_g16).  Row strides 0 = D; padding columns [D, ld) of gx (when not
 * accumulating) and of the bf16 copy are zero-filled.  group_out/skip as in isp_layernorm_fwd (rows = input rows;
 * dropped rows get zero gradient).  Autograd of nn.LayerNorm in dinov2/layers/block.py:92-117, DINOv2.py:533 and of
 * the LayerNorm/ChannelNorm layers of loftup/layers.py:26-58. */
int isp_layernorm_bwd(const void* x, int x_dtype, long ld_x, const void* gy, long ld_gy, const float* gamma, float* gx,
                      long ld_gx, void* gx_bf16, long ld_g16, long rows, int D, float eps, int group_out, int skip,
                      int accumulate, void* stream);

/* ---- LayerNorm affine gradients (a TRAINED LayerNorm: the simple-ViT click encoder, simple_ViT.py:18-155):
 * dgamma[c] += sum_r gy[r][c] * xhat[r][c], dbeta[c] += sum_r gy[r][c]; x fp32 or bf16 (row stride ld_x), gy bf16 (row
 * stride ld_gy); dgamma / dbeta fp32 [D], caller-zeroed (atomics). */
int isp_layernorm_wgrad(const void* x, int x_dtype, long ld_x, const void* gy, long ld_gy, float* dgamma, float* dbeta,
                        long rows, int D, float eps, void* stream);

/* ---- Adjoint of isp_resize_bilinear_ac_nchw_f32 for planar fp32 maps: din [planes,h,w] = R^T dout [planes,H,W].
 * The logits resize of iseg_base_model.py:75-80 under autograd (identity / LiFT-sized upsampler outputs). */
int isp_resize_bilinear_ac_nchw_f32_bwd(const float* dout, float* din, long planes, int h, int w, int H, int W,
                                        void* stream);

/* ---- F.interpolate(mode="bilinear", align_corners=True): basic_upsamplers.py:28-33,
 * iseg_probe_model.py:120-129 (NHWC bf16 feature maps) and iseg_base_model.py:75-80,
 * base_predictor.py:95-97, zoom_in.py:113-118,240-247 (NCHW f32 planes). */
int isp_resize_bilinear_ac_nhwc_bf16(const void* in, void* out, int B, int h, int w, int H, int W, int C, void* stream);
int isp_resize_bilinear_ac_nchw_f32(const float* in, float* out, long planes, int h, int w, int H, int W,
                                    long in_plane_stride, void* stream);

/* ---- F.interpolate nearest / bicubic (align_corners=False) on NHWC bf16:
 * basic_upsamplers.py:18-25,36-42; mode ISP_RESIZE_BILINEAR_AC forwards to the call above. */
#define ISP_RESIZE_NEAREST 0
#define ISP_RESIZE_BILINEAR_AC 1
#define ISP_RESIZE_BICUBIC 2
int isp_resize_nhwc_bf16(const void* in, void* out, int B, int h, int w, int H, int W, int C, int mode, void* stream);

/* ---- click-token injection x[b,(cls)+t,:] += add[b,t,:]: DINOv2.py:516,523. */
int isp_token_add_fwd(void* x, int x_dtype, const void* add, int add_dtype, long B, int T, int D, int x_has_cls,
                      void* stream);

/* ---- FeatUp JBU x2 stage (third-party algorithm, see jbu.hip header; call site
 * core/model/upsamplers/JBUFeatUp.py:18-19).  guidance / proj are fp32, feature maps NHWC bf16.
 * isp_jbu_kernels writes, per output pixel, the COMPOSITE (bicubic-x2 o 7x7 stencil) 8x8 kernel
 * on the low-res source grid: kc [B,GH,GW,8,16] bf16, columns in circular slots (src col & 15).
 * bys [GH,7,8] / bxs [GW,7,16] are the size-only interpolation tables (see jbu.hip);
 * fix0_w / fix3_w are the fix-up MLP weights zero-padded to [64][64] bf16 ([out][in], inputs ordered
 * [kernel(49), guidance(3)]), fix0_b / fix3_b zero-padded to 64 f32 (the MLP runs on MFMA).
 * isp_jbu_apply: out [B,2h,2w,C] = composite kernels applied to src [B,h,w,C] (C % 64 == 0). */
int isp_adaptive_avg_pool_nchw_f32(const float* in, float* out, long planes, int H, int W, int OH, int OW, void* stream);
int isp_jbu_range_proj(const float* guidance, void* proj, const float* w0, const float* b0, const float* w3,
                       const float* b3, int B, int GH, int GW, int exact_f32, const float* drop_hidden, void* stream);
/* exact_f32 != 0: both layers in fp32 on the VALU with the erf GELU, proj [B,GH,GW,32] f32 (checking mode);
 * 0: second layer on f16 MFMA, proj [B,GH,GW,32] IEEE half.  isp_jbu_kernels* take either (proj_f16 = 1 for half).
 * drop_hidden (nullable): train-mode Dropout2d of the frozen stack under the reference's net.train() (trainer.py:214;
 * FeatUp's range_proj / fixup_proj carry Dropout2d(0.1) behind their GELU): per (image, hidden unit) multipliers, 0 or
 * 1/(1-p), [B,32] for isp_jbu_range_proj and [B,64] (49 used) for isp_jbu_kernels*; the caller draws them. */
int isp_jbu_kernels(const void* proj, int proj_f16, const float* guidance, void* kc_bf16, const void* fix0_w, const float* fix0_b,
                    const void* fix3_w, const float* fix3_b, const float* bys, const float* bxs, float range_temp,
                    float sigma_spatial, int B, int GH, int GW, const float* drop_hidden, void* stream);
int isp_jbu_apply(const void* src_nhwc_f16, const void* kc_f16, void* out_nhwc, int B, int h, int w, int C, int out_bf16,
                  void* stream);
/* Last JBU stage fused with the model's bilinear (align_corners) resize to the image size
 * (core/model/iseg_probe_model.py:120-129; FeatUp's x16 map is 16/14 of the image): isp_jbu_blend turns the stage's
 * records [B,GH,GW,8,16] into records of the OUTPUT grid [B,OH,OW,9,16] (OH*8 == GH*7, OW*8 == GW*7), each the bilinear
 * blend of its 2x2 stage records (9 window rows: the lower stage row's window may start one source row later);
 * isp_jbu_apply_resized applies them to the stage's source [B,h,w,C] (GH = 2h) and writes [B,OH,OW,C] directly. */
int isp_jbu_blend(const void* kc_bf16, void* kc9_bf16, int B, int GH, int GW, int OH, int OW, void* stream);
/* isp_jbu_kernels followed by isp_jbu_blend in one launch: the stage's own records are never stored */
int isp_jbu_kernels_resized(const void* proj, int proj_f16, const float* guidance, void* kc9_bf16, const void* fix0_w, const float* fix0_b,
                            const void* fix3_w, const float* fix3_b, const float* bys, const float* bxs, float range_temp,
                            float sigma_spatial, int B, int GH, int GW, int OH, int OW, const float* drop_hidden, void* stream);
int isp_jbu_apply_resized(const void* src_nhwc_f16, const void* kc9_f16, void* out_nhwc, int B, int h, int w, int OH, int OW,
                          int C, int out_bf16, void* stream);
/* Inside the stack everything is IEEE half: kernel records (kc, kc9), the fix-up MLP weights handed to isp_jbu_kernels*,
 * and the feature maps between stages (src / out of isp_jbu_apply*); the stack's bf16 input is converted exactly by
 * isp_bf16_to_f16 and the stage that feeds the seg head writes bf16 (out_bf16 != 0).  n % 8 == 0. */
int isp_bf16_to_f16(const void* in_bf16, void* out_f16, long n, void* stream);

/* Adjoint of isp_jbu_apply w.r.t. the source: gsrc [B,h,w,C] = A(kc)^T gout [B,2h,2w,C] (kc depends on the guidance only).
 * What autograd does for FeatUp's JBU stage when the probe trains through it (models/sbd/dinov2/patch-embed_jbu.py). */
int isp_jbu_apply_bwd(const void* gout_nhwc_bf16, const void* kc_bf16, void* gsrc_nhwc_bf16, int B, int h, int w, int C,
                      void* stream);

/* ---- LoftUp front end: MinMaxScaler statistics (batch-global per-channel min/max,
 * loftup/layers.py:61-71; workspace >= C*B*64*2 floats) and the fused ImplicitFeaturizer
 * (layers.py:107-158, learn_bias, colour feats) + ChannelNorm (layers.py:26-35) producer:
 * image [B,3,H,W] f32 -> out [B,H,W,ldo] bf16, channels [sin(5F) | cos(5F) | colour(3) | 0...]. */
int isp_minmax_nchw_f32(const float* x, float* out_c2, float* workspace, int B, int C, long HW, void* stream);
int isp_loftup_fourier_cn(const float* image, const float* minmax_c2, const float* freqs, const float* bias_sin,
                          const float* bias_cos, const float* gamma, const float* beta, void* out_bf16, int B, int H,
                          int W, int n_freqs, int ldo, float eps, void* stream);
/* same, fp32 output (the fp32 checking mode) */
int isp_loftup_fourier_cn_f32(const float* image, const float* minmax_c2, const float* freqs, const float* bias_sin,
                              const float* bias_cos, const float* gamma, const float* beta, float* out_f32, int B, int H, int W,
                              int n_freqs, int ldo, float eps, void* stream);
/* IEEE-half output (the half-precision inference stream) */
int isp_loftup_fourier_cn_f16(const float* image, const float* minmax_c2, const float* freqs, const float* bias_sin,
                              const float* bias_cos, const float* gamma, const float* beta, void* out_f16, int B, int H, int W,
                              int n_freqs, int ldo, float eps, void* stream);

/* ---- fused ViT MLP branch, in place on the fp32 residual stream x [M][D]:
 *   x += ls2 * (fc2(GELU(fc1(LayerNorm(x)))))      reference dinov2/layers/block.py:92-117 (second branch), mlp.py:34-40,
 *   layer_scale.py:25-26, nn.LayerNorm(eps=1e-6) DINOv2.py:98.  Token-stationary: a workgroup owns 128 rows, their
 *   normalised values stay in registers as MFMA operands, only weights stream through LDS, the [M][HID] hidden map never
 *   exists.  w1 [HID][D] bf16 = fc1.weight * diag(norm2.weight), b1 [HID] f32 = fc1.bias + fc1.weight @ norm2.bias,
 *   w2p [D][HID] bf16 = diag(ls2) * fc2.weight with the hidden axis permuted inside every group of 16 (physical 8g+e holds
 *   logical 4g+e for e < 4, 8+4g+e-4 otherwise, g in {0,1}: the 32x32 accumulator layout of the first product), b2 [D] f32 = ls2 * fc2.bias.
 *   (D, HID) = (384, 1536) (DINOv2-S/14). */
int isp_vit_mlp_fused(float* x, const void* w1, const float* b1, const void* w2p, const float* b2, long M, int D, int HID,
                      float eps, void* stream);
/* The same over the PATCH-token rows only of a [images][rows_per_image][D] stream: rows first_row .. first_row + T - 1 of every
 * image (DINOv2: first_row = 1, rows_per_image = T + 1; row 0 is the class token, DINOv2.py:525-528), tiled per image so that
 * 32 x 1024 patch tokens are exactly 256 workgroups; the class-token rows take the unfused kernels.  w_dtype: ISP_BF16, or
 * ISP_F16 for w1 / w2p (and the kernel's 16-bit operands) in IEEE half -- the trunk's default inference stream. */
int isp_vit_mlp_fused_rows(float* x, const void* w1, const float* b1, const void* w2p, const float* b2, int images,
                           int rows_per_image, int first_row, int T, int D, int HID, float eps, int w_dtype, void* stream);

/* ---- LiFT image pyramid (LiFT.py:70-91,106-112): 3x3 / stride 2 / pad 1 conv to 32 channels (input NCHW f32 with
 * 3 channels, or NHWC bf16 with 32), weights w [32][3][3][cin] f32; relu != 0: eval-BatchNorm folded into w/bias
 * by the caller + ReLU; relu == 0: the raw conv, ahead of isp_bn_train_*.  And F.adaptive_max_pool2d on NHWC bf16. */
int isp_conv3x3_s2_c32(const void* in, int in_is_nchw_f32, int cin, const float* w, const float* bias,
                       void* out_nhwc_bf16, int B, int H, int W, int relu, void* stream);
int isp_adaptive_max_pool_nhwc_bf16(const void* in, void* out, int B, int H, int W, int OH, int OW, int C, void* stream);

/* ---- BaseClassifierHead.classifier (1x1 conv C->1), heads/base_head.py:15.
 * x [M,C] NHWC bf16, weight [C] f32 -> out [M] f32. */
int isp_classifier_fwd(const void* x_nhwc_bf16, const float* weight, float bias, float* out, long M, int C,
                       void* stream);

/* ---- predictor fusion: AddHorizontalFlip.inv_transform + SigmoidForPred.inv_transform
 * (core/inference/transforms/flip.py:32-36, base_transform.py:39).  logits [2n,1,H,W] (or [n,..]
 * when with_flip=0) f32 -> probs [n,1,H,W] f32. */
int isp_fuse_flip_sigmoid(const float* logits, float* probs, long n, int H, int W, int with_flip, void* stream);

/* ---- backward of the trainable tail (loss.backward(), core/training/trainer.py:224, for clicks
 * injected after the frozen backbone): weight gradients as a pixel-reduction ("TN") GEMM with an
 * optional implicit 3x3 tap shift of Q (conv wgrad: call once per tap with out + tap*C), fp32
 * atomics into a caller-zeroed buffer; ReLU mask + bias gradient; classifier backward; adjoint of
 * the align_corners bilinear resize.  Activation gradients of 3x3 convs reuse
 * isp_conv3x3_nhwc_bf16 with the 180-degree rotated, transposed weights. */
int isp_tn_gemm_bf16_atomic(const void* P, long ldp, const void* Q, long ldq, float* out, long ldo, long M, int N, int J,
                            int shift_H, int shift_W, int shift_dy, int shift_dx, int splits, void* stream);
/* dW[n][(ky*3+kx)*C + c] += sum_pixels g[b,y,x,n] * x[b,y+ky-1,x+kx-1,c]: the 3x3 conv weight gradient with all nine
 * taps formed from one staged input patch (g [B,H,W,N], x [B,H,W,C] bf16 NHWC; dW fp32 [N][9*C], caller-zeroed,
 * fp32 atomics).  Autograd of ConvModule.conv.weight, heads/conv_heads.py:51-73 under trainer.py:219-226. */
int isp_conv3x3_wgrad_bf16_atomic(const void* g, const void* x, float* dw, int B, int H, int W, int C, int N, void* stream);
/* ---- train-mode BatchNorm2d of the frozen upsamplers.  The reference's `self.net.train()` (core/training/trainer.py:214,
 * 431) puts EVERY module in training mode, including the BatchNorm2d layers of the frozen LiFT (LiFT.py:19-24,71-76,88)
 * and LoftUp (loftup/loftup.py:58,63): they normalise with batch statistics and update their running ones.
 * x, y, dy, dx: bf16 [M = B*H*W][C] (NHWC); C % 8 == 0, C <= 2048; sums / gsums: fp32 [2*C], caller-zeroed.
 *   isp_bn_train_stats  sums = {sum_m x, sum_m x^2}
 *   isp_bn_train_apply  y = act((x - mean) * rsqrt(var + eps) * gamma + beta) from those sums (biased variance);
 *                       new_mean/new_var (nullable, [c_real]) = (1-momentum)*running + momentum*{mean, unbiased var}
 *   isp_bn_train_bwd    dx of the same op for dy taken AFTER the ReLU whose output is y (y nullable: no ReLU):
 *                       g = dy*[y>0]; dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)); gsums = {sum g, sum g*xhat} */
int isp_bn_train_stats(const void* x, float* sums, long M, int C, void* stream);
int isp_bn_train_apply(const void* x, const float* sums, const float* gamma, const float* beta, void* y, long M, int C,
                       float eps, int relu, const float* running_mean, const float* running_var, float* new_mean,
                       float* new_var, int c_real, float momentum, void* stream);
int isp_bn_train_bwd(const void* dy, const void* x, const void* y, const float* sums, const float* gamma, float* gsums,
                     void* dx, long M, int C, float eps, void* stream);
int isp_relu_mask_colsum(const void* dy, const void* y, void* g, float* colsum, long M, int N, void* stream);
/* dx_colsum (nullable, [C], caller-zeroed): += column sums of dx = bias gradient of the conv that produced x.
 * relu_mask != 0: x is the output of a conv+ReLU layer and dx = (x > 0) * g * w (that layer's ReLU backward fused in);
 * relu_mask == 0: x is a signed feature map (SimpleClassifierHead, heads/conv_heads.py:10-24; num_layers == 0) and
 * dx = g * w. */
int isp_classifier_bwd(const float* grad_logits, const void* x, const float* w, void* dx, float* dw, float* db,
                       float* dx_colsum, long M, int C, int relu_mask, void* stream);
int isp_resize_bilinear_ac_nhwc_bwd(const void* dout, void* din, int B, int h, int w, int H, int W, int C, void* stream);

/* ---- layout converters between the plugin API (NCHW f32) and the kernels (NHWC bf16).
 * The f32 source is addressed in[b*sb + c*sc + p*sp] so permuted views need no copy. */
int isp_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int C, long HW, void* stream);
int isp_nchw_f32_to_nhwc_bf16(const float* in, void* out, int B, int C, long HW, long stride_b, long stride_c,
                              long stride_p, void* stream);

/* ---- Device-side robot user.  One call = Clicker._get_next_click (core/inference/
 * clicker.py:58-91) + utils.get_iou (core/inference/utils.py:107-120) for one prediction, on masks resident in HBM:
 * pred / gt / not_ignore (nullable = all ones) / not_clicked are uint8 [H,W] (0/1).  out[8] int32 (device):
 * {is_positive, row, col, max squared interior distance of the FN region, of the FP region, |pred&gt&ni|, |(pred|gt)&ni|, 0}.
 * Exact integer EDT of the 1-pixel zero-padded masks (what cv2.distanceTransform(DIST_L2, maskSize 0) computes before
 * its final sqrt), already-clicked pixels zeroed, larger maximum wins, first maximum in row-major order.
 * workspace: isp_robot_click_workspace_bytes(H, W) bytes.  not_clicked is NOT modified (the caller records clicks). */
long isp_robot_click_workspace_bytes(int H, int W);
int isp_robot_click(const void* pred, const void* gt, const void* not_ignore, const void* not_clicked, int H, int W,
                    void* workspace, int* out, void* stream);
/* mask[i] = probs[i] > thr (uint8), evaluation.py:74 */
int isp_threshold_u8(const float* probs, void* mask, float thr, long n, void* stream);

/* ---- Train-time click simulation on the device: get_next_points, core/training/
 * trainer.py:575-618.  pred [B,1,H,W] f32 probabilities, gt [B,1,H,W] f32 (> 0.5 = object), points [B,2P,3] f32
 * updated IN PLACE at slot (P - click_indx) for a positive / (2P - click_indx) for a negative click with
 * (row, col, click_indx); rand32 [B] uint32 (device) = the caller's uniform draw, index = (rand32 * n) >> 32 among the
 * n pixels with dt > max/2 in row-major order (the reference: np.random.randint(0, n)).  dt = OpenCV's 5x5 chamfer
 * DIST_L2 transform (16.16 fixed point, costs 1 / 1.4 / 2.1969) of the zero-padded FN / FP masks.  H + 2 <= 1024.
 * workspace: isp_next_points_workspace_bytes(B, H, W) bytes.  No host synchronisation. */
long isp_next_points_workspace_bytes(int B, int H, int W);
int isp_next_points(const float* pred, const float* gt, float* points, const unsigned* rand32, int B, int H, int W, int P,
                    int click_indx, float pred_thresh, void* workspace, void* stream);

/* ---- FeatUp JBU stage in plain fp32, as the published algorithm states it (the fp32 checking mode and an on-device
 * cross-check of the composite-kernel formulation above): per-pixel 49-tap kernels k [B,GH,GW,49] from proj [B,GH,GW,32]
 * and the pooled guidance [B,3,GH,GW] (fix-up MLP weights fp32: fixup_proj.0 [49,52], fixup_proj.3 [49,49]);
 * F.interpolate(bicubic, align_corners=False) x2 on NHWC fp32; the reflect-padded 7x7 adaptive convolution. */
int isp_jbu_kernels_f32(const float* proj, const float* guidance, float* k_out, const float* fix0_w, const float* fix0_b,
                        const float* fix3_w, const float* fix3_b, float range_temp, float sigma_spatial, int B, int GH, int GW,
                        void* stream);
int isp_bicubic_x2_nhwc_f32(const float* src, float* out, int B, int h, int w, int C, void* stream);
int isp_adaptive_conv7_nhwc_f32(const float* hr, const float* k49, float* out, int B, int GH, int GW, int C, void* stream);

/* ---- fp32-accurate products on the bf16 engine (core/model/precise.py; the "logits within 1e-3 fp32" gate of the
 * reference comparison).  isp_split_bf16x3 writes an fp32 [rows, K] matrix (row stride ld_in) as bf16 [rows, 3*Kpad]:
 * activations layout [hi | hi | lo] (weights_layout = 0) or weights layout [hi | lo | hi] (1), hi = bf16(v),
 * lo = bf16(v - hi), v = scale * act(x) with act 0 none / 1 ReLU / 2 GELU(erf) / 3 QuickGELU; columns K..Kpad are zero.  A GEMM or conv
 * over the tripled depth then accumulates hi.whi + hi.wlo + lo.whi in fp32.  isp_softmax_rows_f32: in-place softmax over
 * the first `cols` entries of each fp32 row (the rest of the row, up to ld, is zeroed). */
int isp_split_bf16x3(const float* x, long ld_in, void* out_bf16, long rows, int K, int Kpad, int weights_layout, int act,
                     float scale, void* stream);
int isp_softmax_rows_f32(float* x, long rows, int cols, long ld, void* stream);
/* Self-attention of the ViT trunk in exact fp32 (the checking mode's form of dinov2/layers/attention.py:54-71):
 * out [B*L, heads*64] = softmax((q * scale) k^T) v per (batch, head) from the packed fp32 qkv [B*L, 3*heads*64], head_dim 64,
 * one launch (flash form on the f32-input MFMA, exact fp32 multiply-adds, fp32 online softmax). */
int isp_attention_packed_f32(const float* qkv, float* out, int B, int L, int heads, float scale, void* stream);

/* ---- On-box roofline probes (diagnostics; tools/peaks.py): a register-resident v_mfma_f32_16x16x32_bf16 loop
 * (blocks x 4 waves x iters x 16 MFMAs; operands read once from a 64 Ki-element bf16 seed: zeros vs random bits show the
 * clock the chip holds) and a float4 copy of `bytes` bytes. */
int isp_probe_mfma_bf16(const void* seed_bf16_64k, float* sink, int blocks, int iters, void* stream);
/* same loop on v_mfma_f32_32x32x16_bf16: blocks x 4 waves x iters x 8 MFMAs of 32x32x16 */
int isp_probe_mfma_bf16_32x32(const void* seed_bf16_64k, float* sink, int blocks, int iters, void* stream);
int isp_probe_copy(const void* src, void* dst, long bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISEGPROBE_HIP_H */
