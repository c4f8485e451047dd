/*
 * dtk.h -- C-ABI of libdtk.so: the MI355X (gfx950) device programs behind DINO-Tracker's per-video
 * inference hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise.
 *
 * The reference (/root/reference, pure PyTorch) has no FFI of its own: the boundary it exposes is the Python API
 * of models/tracker.py, models/model_inference.py, models/extractor.py.  Each entry point
 * below names the reference call site(s) it replaces (file:line in /root/reference); the Python classes in
 * dino_tracker_amd/ keep the reference's names and signatures and call these through ctypes
 * (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Conventions
 *   - return 0 on success, negative DTK_E_* on failure; dtk_last_error() gives a message (thread-local).
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream); no host sync inside unless
 *     stated; the caller owns every buffer, nothing is allocated or retained by the library.
 *   - feature volume layout ("token-major"): F[t][cell][c], cell = row*pw + col, c contiguous, fp32.
 *   - pixel coordinates are (x, y) at model resolution; cell (row, col) centre = (stride*col + patch/2,
 *     stride*row + patch/2)  (models/networks/tracker_head.py:72-81).
 */
#ifndef DTK_H_
#define DTK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTK_OK 0
#define DTK_E_INVALID (-1)   /* bad argument / unsupported size */
#define DTK_E_HIP (-2)       /* a HIP runtime call or kernel launch failed */
#define DTK_E_WORKSPACE (-3) /* workspace too small */

/* geometry of one video at model resolution; ph/pw = 1 + (dim - patch) / stride (models/extractor.py:171-181) */
typedef struct dtk_geom {
    int32_t T;        /* frames */
    int32_t C;        /* feature width (384 ViT-S, 768 ViT-B, 1024 ViT-L) */
    int32_t ph, pw;   /* token grid */
    int32_t video_h, video_w;
    int32_t patch, stride;
    float radius;     /* soft-argmax disk radius in px (tracker_head.py:47, 35) */
} dtk_geom;

/* TrackerHead.cnn_refiner parameters AFTER NormalizedConv2d's W / sum(W) (conv_norm.py:34-46), which
 * dtk_head_prepare computes on device.  Layout: w1[16][9], b1[16], w2[16][9] (w2[ch][tap]), b2[1]; tap = 3*ky+kx. */
#define DTK_HEAD_HIDDEN 16
#define DTK_HEAD_PARAMS (16 * 9 + 16 + 16 * 9 + 1)

int dtk_version(void);
const char* dtk_last_error(void);

/* ---- per-kernel timing (used by bench.py for the roofline figures) -----------------------------------------
 * dtk_profile_enable(1) makes every kernel launch of this library be bracketed by hipEvents on its stream;
 * dtk_profile_collect() waits for them and aggregates per kernel name; *_name/_ms/_launches read entry i. */
int dtk_profile_enable(int on);
int dtk_profile_collect(void);
const char* dtk_profile_name(int i);
double dtk_profile_ms(int i);
long long dtk_profile_launches(int i);

/* ---- feature volume ------------------------------------------------------------------------------------ */
/* Reference layout [T][C][ph][pw] (dino_embed_video.pt, models/tracker.py:65-71) -> token-major [T][HW][C]
 * plus per-cell L2 norms [T][HW] (the `.norm(dim=1)` of models/tracker.py:162, hoisted to once per video). */
int dtk_pack_features(const float* chw, float* thwc, float* norms, int T, int C, int HW, void* stream);
/* inverse layout conversion (for callers that read `.refined_features` as T x C x h x w) */
int dtk_unpack_features(const float* thwc, float* chw, int T, int C, int HW, void* stream);
/* norms only (after refinement wrote thwc in place) */
int dtk_feature_norms(const float* thwc, float* norms, int T, int C, int HW, void* stream);

/* ---- P1: DINOv2 ViT encoder as driven by VitExtractor (models/extractor.py:23-150, utils.py:33-72) -----------------
 * Host-side structs of DEVICE pointers.  Matrix weights are 16-bit in nn.Linear layout [out][in], in the model's OPERAND
 * TYPE: IEEE fp16 by default (round 3: the same MFMA rate as bf16 with 8x less operand rounding -- round 2's end-to-end
 * error from the video was the bf16 operands), bf16 with DTK_VIT_BF16; vectors are fp32.
 * Upstream facebookresearch/dinov2 parameter names are given for each field (blocks.{i}.*). */
typedef struct dtk_vit_layer {
    const float *ln1_w, *ln1_b;   /* norm1.weight / .bias */
    const void* qkv_w;            /* attn.qkv.weight  16-bit [3D][D] */
    const float* qkv_b;           /* attn.qkv.bias    [3D] */
    const void* proj_w;           /* attn.proj.weight 16-bit [D][D] */
    const float* proj_b;          /* attn.proj.bias */
    const float* ls1;             /* ls1.gamma */
    const float *ln2_w, *ln2_b;   /* norm2.* */
    const void* fc1_w;            /* mlp.fc1.weight 16-bit [4D][D] */
    const float* fc1_b;
    const void* fc2_w;            /* mlp.fc2.weight 16-bit [D][4D] */
    const float* fc2_b;
    const float* ls2;             /* ls2.gamma */
    /* Round 6 -- the ESCALATED precision of a block (csrc/vit_split.h): when qkv_w_lo is not NULL, this block runs every matrix
     * product on split operands (x = hi + lo, three MFMAs per product, fp32-grade results at ~3x the matrix work): the four
     * `*_w` pointers above are then the HI planes and these the LO planes of  w_scale * W  (lo = T(w_scale W - hi); w_scale a
     * power of two -- 2^8 for fp16 so that the lo halves stay normal numbers, 1 for bf16), all four or none.  Blocks can be
     * escalated individually; a model whose operand type is bf16 keeps 16 significant bits this way at fp32's range. */
    const void *qkv_w_lo, *proj_w_lo, *fc1_w_lo, *fc2_w_lo;
    float w_scale;
} dtk_vit_layer;

typedef struct dtk_vit_model {
    int32_t D, heads;             /* d_head = D / heads must be 64 (ViT-S/B/L) */
    int32_t depth;                /* number of blocks to run = hooked layer + 1 (models/extractor.py:137-150) */
    int32_t patch, stride;        /* 14, 7 (models/extractor.py:41-55) */
    float ln_eps;                 /* 1e-6 */
    int32_t flags;                /* 0 or a combination of the DTK_VIT_* bits below */
    const float* patch_w;         /* patch_embed.proj.weight fp32 [D][3][patch][patch] */
    const float* patch_b;         /* patch_embed.proj.bias */
    const float* cls_pos;         /* cls_token + pos_embed[0]  [D] */
    const float* pos;             /* interpolated patch position encoding [ph*pw][D] (models/extractor.py:57-85) */
    const float* mean_std;        /* ImageNet mean[3], std[3] (utils.py:46) */
    const dtk_vit_layer* layers;  /* HOST array of `depth` entries */
    int32_t frame_batch;          /* frames per pass of the encoder (workspace grows with it: 87 MB per frame at 854 x 476, ViT-S); 0 = the library's default (90) */
    int32_t* overflow;            /* DEVICE word or NULL; OR-ed with 1 when a residual update (projection / MLP output) reached
                                   * the fp16 limit 65504 or is not finite (every token of every frame: the LayerNorm that
                                   * applies the update checks it), with 2 / 4 when a value the QKV / fc1 GEMM STORES (Q after
                                   * its scale, K, V / the MLP hidden AFTER the GELU) did: tracked inside those GEMMs' epilogues for every value of every
                                   * frame (round 5; DTK_VIT_CHECK_RANGE adds a scan of the stored tensors, the tests'
                                   * cross-check).  The caller zeroes it.  fp16 activations
                                   * SATURATE (FP16_OVFL mode) instead of becoming inf; a non-zero word means the features
                                   * are not trustworthy and the model should be run with DTK_VIT_BF16. */
    /* Round 6 -- the multi-layer form of get_feature_from_input (models/extractor.py:142-149: the MEAN over the requested layers of
     * the block outputs) in ONE pass: after every block l with bit l of tap_mask set, tap_out [n][1 + ph*pw][D] += tap_scale x (the
     * residual stream after block l).  The caller zeroes tap_out and sets tap_scale = 1 / #layers.  NULL / 0: off. */
    float* tap_out;
    uint64_t tap_mask;
    float tap_scale;
} dtk_vit_model;

#define DTK_VIT_TILED_GEMMS 1   /* run the K = 384 GEMMs on the tiled kernel too (cross-check in the tests) */
#define DTK_VIT_BF16 2          /* operand type bf16 instead of fp16 (weights must then be bf16) */
#define DTK_VIT_CHECK_RANGE 4   /* scan Q / K / V^T and the MLP hidden of every block for saturated values (costs a pass) */
#define DTK_VIT_ATTENTION_V2 8  /* attention on the round-2/3 kernel (16 waves per CU x 32 queries) instead of the one-wave-per-SIMD
                                 * kernel of round 4: the cross-check path of the tests */
#define DTK_VIT_GEMM_WS_V1 16   /* the K = 384 weight-stationary GEMMs in their round 1-3 form (A / B measurement, cross-check) */
#define DTK_VIT_ATTENTION_V4 32 /* attention on the round-4/5 kernel (one wave per SIMD, 64 queries per wave: csrc/vit_attention4.h) instead of
                                 * round 6's 128 queries per wave (csrc/vit_attention6.h): A / B measurement, cross-check */
#define DTK_VIT_NO_LN_FUSION 128 /* D = 384: keep the LayerNorm of the next block a launch of its own instead of running it inside fc2's
                                  * epilogue (A / B measurement, cross-check: the results are bit-identical) */
#define DTK_VIT_GEMM_WIDE_V1 64 /* the LDS-DMA GEMMs (256 x 256 tiles of D = 768 / 1024, fc2 of D = 384, the split-operand GEMMs) in their
                                 * round 4-5 form: 8-byte stores straight from the D tiles instead of whole rows through LDS, fragments read
                                 * at the top of every k-step (A / B measurement, cross-check: the results are bit-identical) */
#define DTK_OPERAND_F16 0
#define DTK_OPERAND_BF16 1
#define DTK_OPERAND_ATTENTION_V2 0x100  /* OR-ed into dtk_vit_attention's operand_type: the same selection for the stand-alone stage */
#define DTK_OPERAND_ATTENTION_V4 0x2000 /* the same selection as DTK_VIT_ATTENTION_V4 for the stand-alone stage */
#define DTK_OPERAND_ATTENTION_V5 0x200  /* stand-alone stage only: the round-5 EXPERIMENT kernel, two waves per SIMD alternating matrix /
                                         * vector phases (csrc/vit_attention5.h); measured against the library's kernel by
                                         * scripts/attn_ab.py, not used by dtk_vit_forward */
#define DTK_OPERAND_ATTENTION_V5_INPHASE 0x400  /* with ..._V5: both wave halves in phase (the experiment's ablation); 0x800 / 0x1000 with
                                                 * ..._V5: micro-benchmark ablations WITHOUT meaningful output (no vector / no matrix work) */

/* frames [n][3][video_h][video_w] fp32 in [0,1] -> block output of layer depth-1 (before the final norm):
 * tokens_out [n][1 + ph*pw][D] (CLS first; what get_feature_from_input returns) and/or
 * feat_out   [n][ph*pw][D]     (CLS dropped: the token-major feature volume of this library) and/or
 * qkv_out    [n][1 + ph*pw][3D] the output of blocks[depth-1].attn.qkv (the tensor the reference's qkv hook records,
 *            models/extractor.py:107-118; the key / query / value facets are reshapes of it, :245-267).
 * Any of them may be NULL (not all). */
size_t dtk_vit_workspace_bytes(const dtk_vit_model* m, int video_h, int video_w, int frames);
int dtk_vit_forward(const dtk_vit_model* m, const float* frames, int nframes, int video_h, int video_w,
                    float* tokens_out, float* feat_out, float* qkv_out, void* workspace, size_t workspace_bytes,
                    void* stream);

/* The multi-head self-attention stage of a block on its own, d_head = 64 (what runs between the QKV and the projection
 * GEMMs inside dtk_vit_forward; upstream Attention.forward, hooked by models/extractor.py:101-104).  Operands in
 * `operand_type` (DTK_OPERAND_F16 / DTK_OPERAND_BF16):
 *   q  [frames][heads][Sp][64]   queries, ALREADY multiplied by log2(e) / sqrt(64) (the softmax runs in the exp2 domain)
 *   k  [frames][heads][Sp][64]   keys; rows S .. Sp-1 must be finite (zero)
 *   vt [frames][heads][64][Sp]   values, transposed; columns S .. Sp-1 must be finite (zero)
 *   out[frames][S][heads*64]     softmax(q k^T) v, heads concatenated (the input of attn.proj)
 * Sp = S rounded up to a multiple of 64 (dtk_vit_forward uses 128). */
int dtk_vit_attention(const void* q, const void* k, const void* vt, void* out, int frames, int heads, int S, int Sp,
                      int operand_type, void* stream);
/* The same stage on SPLIT operands (the escalated precision of a block, dtk_vit_layer.qkv_w_lo): every tensor as hi / lo planes
 * of `operand_type` in the layouts above (x = hi + lo); scores, probabilities and both products carry ~22 (fp16) / ~16 (bf16)
 * significant bits. */
int dtk_vit_attention_split(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                            const void* vt_lo, void* out_hi, void* out_lo, int frames, int heads, int S, int Sp, int operand_type,
                            void* stream);

/* ---- P2: Delta-DINO refinement (models/tracker.py:113-135; models/networks/delta_dino.py:53-61;
 *      models/utils.py:7-45), fp32-grade on the fp16 MFMA (operands split into hi + lo halves, 3 products) ----------
 * dtk_delta_dino_pack: layer l in 0..3 (state-dict keys layers.{4l}.{weight,bias} = conv [Cout][Cin][5][5] and
 *   layers.{4l+1}.{weight,bias,running_mean,running_var} = BatchNorm2d, eval mode) -> kernel layout
 *   packed = Wk[25][CinP][CoutP] | scale[CoutP] | shift[CoutP] | split-fp16 weight planes (hi, lo; 2^8 w) in the
 *   tile order of the convolution kernels  (dtk_delta_dino_packed_floats floats in total; opaque to the caller).
 * dtk_delta_dino_refine: for frames t0 .. t0+nframes-1: out[t] = dino[t] + align(CNN(video[t])), token-major, plus
 *   per-cell norms (may be NULL).  video is [T][3][video_h][video_w] fp32 in [0,1] (data/data_utils.py:79-104),
 *   packed = 4 device pointers (host array).  BlurPool = antialiased_cnns.BlurPool(stride 2): reflect-pad (1,2,1,2),
 *   4x4 binomial. */
size_t dtk_delta_dino_packed_floats(int layer, int C);
int dtk_delta_dino_pack(int layer, int C, const float* w, const float* bias, const float* bn_w, const float* bn_b,
                        const float* bn_mean, const float* bn_var, float eps, float* packed, void* stream);
size_t dtk_delta_dino_workspace_bytes(const dtk_geom* g);
int dtk_delta_dino_refine(const dtk_geom* g, const float* video, const float* dino, const float* const* packed,
                          float* out, float* norms, int t0, int nframes, void* workspace, size_t workspace_bytes,
                          void* stream);
/* The same with the operand precision of the 5x5 convolutions of layers 2-4 chosen by the caller:
 *   DTK_DD_SPLIT  every fp32 operand as hi + lo fp16 halves, three MFMA products per term (fp32-grade: 3e-5 of the reference)
 *   DTK_DD_FP16   the hi halves only: plain fp16 operands, fp32 accumulation, one product per term
 * DEFAULTS DIFFER BY ENTRY POINT, on purpose: dtk_delta_dino_refine (above, the round 1-3 entry) always means DTK_DD_SPLIT, so a
 * C caller that never heard of the mode keeps the fp32-grade result; the Python mirror (dino_tracker_amd/delta_dino.py) calls
 * THIS entry and defaults to DTK_DD_FP16 since round 4 (the end-to-end position error is unchanged to 1e-5 px behind the fp16
 * ViT operands, profiles/r04_e2e_error_p2_operands.json; DTK_P2_OPERANDS=split or DeltaDINO.conv_operands = "split" restores the
 * other).  tests/test_gpu_p2.py pins both modes against the reference-written golden, each with its own tolerance. */
#define DTK_DD_SPLIT 0
#define DTK_DD_FP16 1
int dtk_delta_dino_refine_mode(const dtk_geom* g, const float* video, const float* dino, const float* const* packed,
                               float* out, float* norms, int t0, int nframes, int operands, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- K8: bilinear point sampling -------------------------------------------------------------------------
 * Tracker.sample_embeddings (models/tracker.py:96-111 -> utils.py:75-101) for integral frame indices:
 * out[b][:] = bilinear(F[t_idx[b]], (xy[b] - patch/2) / stride), border clamp, align_corners.
 * out row b is written at out + out_row[b]*C if out_row != NULL else out + b*C. */
int dtk_sample_points(const dtk_geom* g, const float* feat, const float* xy, const int32_t* t_idx,
                      const int32_t* out_row, float* out, int B, void* stream);

/* Literal utils.bilinear_interpolate_video (utils.py:75-101; what Tracker.sample_embeddings calls, models/tracker.py:96-111)
 * on a token-major volume feat[T][ph*pw][C]: pts[B][3] = (x, y, t) ALREADY in [-1, 1] (grid_sample coordinates),
 * trilinear, align_corners=True, border padding -> out[B][C].  No patch / stride enters. */
int dtk_sample_grid(const float* feat, int T, int C, int ph, int pw, const float* pts, float* out, int B, void* stream);

/* ---- TrackerHead parameters ---------------------------------------------------------------------------- */
/* raw state-dict tensors (cnn_refiner.0.weight [16,1,3,3], .0.bias [16], .2.weight [1,16,3,3], .2.bias [1])
 * -> normalised packed parameters head[DTK_HEAD_PARAMS] (conv_norm.py:34-46). */
int dtk_head_prepare(const float* w1, const float* b1, const float* w2, const float* b2, float* head, void* stream);

/* TrackerHead.forward alone (models/networks/tracker_head.py:107-121) on B ReLU'd cost volumes maps[B][ph*pw]:
 * out[b] = normalised (x, y) in [-1,1] (normalized != 0) or pixels. */
int dtk_head_forward(const dtk_geom* g, const float* head, const float* maps, float* out_xy, int B, int normalized,
                     void* stream);

/* TrackerHead.forward / backward of the TRAINING step (tracker_head.py:68-121 under autograd).  `head` = the packed parameters of
 * dtk_head_prepare (305 floats: w1[16][9] | b1[16] | w2[16][9] | b2, weights already normalised W / sum W).
 * dtk_head_forward_train = dtk_head_forward + stats[b][4] = (arg-max cell as int bits, softmax maximum, partition sum, disk
 * mass before the zero-mass fallback).  dtk_head_backward: grad_out[b][2] -> dmaps[b][ph*pw] (gradient with respect to the
 * input maps; the caller ZEROES it: only the 15 x 15 window around each arg-max is written) and dhead_partial[b][305] (per-map
 * parameter gradients, same packing; the caller sums over b).  Exact when no map's fallback fired (stats[b][3] >= 1e-8): then
 * the output does not depend on logits outside the disk and the backward is local to the window; the caller must take another
 * route for a batch in which a fallback fired.  Needs radius / stride <= 5. */
int dtk_head_forward_train(const dtk_geom* g, const float* head, const float* maps, float* out_xy, float* stats, int B,
                           int normalized, void* stream);
int dtk_head_backward(const dtk_geom* g, const float* head, const float* maps, const float* stats, const float* grad_out,
                      float* dmaps, float* dhead_partial, int B, int normalized, void* stream);

/* The split-fp16 implicit-GEMM 5 x 5 convolution as a stand-alone operator: forward and data gradient of the Delta-DINO layers
 * in the TRAINING step (models/networks/delta_dino.py:29-44 under autograd), fp32-grade (operands as hi + lo fp16 planes, three
 * matrix-core products per term), stride 1, "same" size, dilation 1 or 2, reflect or zero padding.  Cin a multiple of 16.
 *   dtk_conv_split_pack     w [Cout][Cin][5][5] -> the kernel's weight planes (dtk_conv_split_weight_halves(Cin, Cout) fp16 values
 *                           each).  flip_transpose: w is [Cin][Cout][5][5] and taps are read reversed -- the operator of the data
 *                           gradient with respect to the forward's input.
 *   dtk_conv_split_input    x [N][C][H][W] fp32 times *scale (device scalar or NULL) -> hi / lo planes [N][H+2b][W+2b][C] with a
 *                           ring of b zero pixels.  C a multiple of 8.
 *   dtk_conv_split_run      planes [N][H][W][Cin] -> out [N][H][W][Cout] fp32.  fp16_only: use the hi halves only (plain fp16
 *                           operands, 2^-11 relative per operand, a third of the matrix-core work) -- the "fp16" training mode; the
 *                           same switch exists on dtk_conv_wgrad_split.
 *   dtk_conv_split_output   y [N][H+2b][W+2b][C] -> [N][C][H][W] divided by *scale; reflect_fold adds the ring back onto the image
 *                           with the adjoint of the reflect padding (b = 2 * dilation: the data gradient of a reflect-padded
 *                           convolution is the zero-padded convolution of dY over the padded domain, folded). */
size_t dtk_conv_split_weight_halves(int Cin, int Cout);
int dtk_conv_split_pack(const float* w, int Cin, int Cout, int flip_transpose, void* Wh, void* Wl, void* stream);
int dtk_conv_split_input(const float* x, int N, int C, int H, int W, int border, const float* scale, void* hi, void* lo,
                         void* stream);
int dtk_conv_split_run(const void* in_hi, const void* in_lo, const void* Wh, const void* Wl, float* out_nhwc, int N, int H, int W,
                       int Cin, int Cout, int dilation, int zero_pad, int fp16_only, void* stream);
int dtk_conv_split_output(const float* y_nhwc, int N, int C, int H, int W, int border, int reflect_fold, const float* scale,
                          float* out_nchw, void* stream);

/* Weight gradient of the same convolution without an unfolded operand: dw [Cout][Cin][5][5] (overwritten) = sum over frames and
 * pixels of dy [N][Cout][H][W] (times *scale_dy, a power of two, undone on output) and the padded x [N][Cin][H][W]; both operands
 * fp32 NCHW as autograd holds them, split into fp16 halves while staged (fp32-grade).  workspace: the per-workgroup partial sums
 * (dtk_conv_wgrad_split_workspace_bytes), reduced by a second kernel -- no atomics, deterministic. */
size_t dtk_conv_wgrad_split_workspace_bytes(int N, int Cin, int Cout, int H, int W, int dilation);
int dtk_conv_wgrad_split(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, int dilation,
                         int reflect_pad, const float* scale_dy, int fp16_only, void* workspace, size_t workspace_bytes,
                         void* stream);

/* The contrastive (InfoNCE) terms of the training loss (dino_tracker.py:141-243, :327-343) without the affinity tensors:
 * problem q < Q = B anchor embeddings a[q][i][C] against all n cells of frame fidx[q] of fe[F][C][n]:
 *     s[i][j] = cos(a_i, f_j) (clamped at 1e-8 like the reference),   lse[q][i] = log sum_j exp(s[i][j] / temp)
 * so that the reference's term -log(exp(cos(a_i, b_i) / temp) / sum_j exp(s[i][j] / temp)) = lse - cos(a_i, b_i) / temp.
 * forward writes lse and keeps for backward: nf[F][n] = |f_j|, na[Q][B] = |a_i|, S[Q][B][np] = the cosines (np = n rounded up
 * to a multiple of 4); fet [F][n][C] is scratch.  backward: g[Q][B] = dL/dlse -> da[Q][B][C] (written) and dfe[F][C][n]
 * (ZEROED and written: the sum over the problems of each frame; float atomics, so the last bits depend on the order). */
size_t dtk_contrastive_workspace_bytes(int Q, int B, int C, int n);
int dtk_contrastive_forward(const float* fe, const float* a, const int32_t* fidx, float temp, int Q, int B, int C, int n, int F,
                            float* fet, float* nf, float* na, float* S, float* lse, void* stream);
int dtk_contrastive_backward(const float* fe, const float* a, const int32_t* fidx, float temp, int Q, int B, int C, int n, int F,
                             const float* nf, const float* na, const float* S, const float* lse, const float* g, float* da,
                             float* dfe, void* workspace, size_t workspace_bytes, void* stream);

/* The embedding regularisers of the training loss (dino_tracker.py:128-139): out2 = (mean | |x| / |raw| - 1 |, mean | cos(x, raw) - 1 |)
 * over the F * n cells of x, raw [F][C][n]; cell_sums [3][F * n] keeps the per-cell sums for dtk_emb_reg_backward, which writes
 * dx [F][C][n] for the upstream gradients grad_out2 (device, two floats). */
int dtk_emb_reg_forward(const float* x, const float* raw, int F, int C, int n, float* cell_sums, float* out2, void* stream);
int dtk_emb_reg_backward(const float* x, const float* raw, const float* cell_sums, const float* grad_out2, int F, int C, int n,
                         float* dx, void* stream);

/* Backward of the cosine maps (models/tracker.py:158-173 under autograd) behind dtk_head_backward: maps[b] = relu'd cosine map
 * of emb[b] against frame tgt[b] (dtk_corr_maps with relu = 1), dmaps[b] its gradient (non-zero only on the 15 x 15 window around
 * the arg-max cell stats[b][0], as dtk_head_backward leaves it).  demb[b][C] is written; dfeat[T][ph*pw][C] (token-major, ZEROED
 * by the caller) receives atomic adds.  C <= 1024. */
int dtk_corr_window_backward(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* tgt,
                             const float* maps, const float* dmaps, const float* stats, float* demb, float* dfeat, int B,
                             void* stream);

/* NormalizedConv2d.forward as a stand-alone layer (models/networks/conv_norm.py:34-46): x[B][Cin][H][W],
 * w[Cout][Cin][k][k] (RAW weights: the per-kernel W / sum(W) is applied inside), bias[Cout] or NULL -> y[B][Cout][H][W];
 * stride 1, zero padding k/2, k odd.  (The tracker path runs these layers fused inside dtk_track / dtk_head_forward.) */
int dtk_normalized_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H,
                          int W, int k, void* stream);

/* Tracker.get_corr_maps_for_frame_set (models/tracker.py:158-169): cosine maps of M sources against their target
 * frames, maps[m][ph*pw] fp32; relu != 0 applies cmap_relu (:173).  snorm_scratch: M floats. */
int dtk_corr_maps(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* src_row,
                  const int32_t* tgt, float* maps, float* snorm_scratch, int M, int relu, void* stream);

/* ---- K9-K13: track sources into target frames ------------------------------------------------------------
 * For each source m < M:  s = emb[src_row[m]] (row of an [R][C] fp32 matrix), a = tgt[m]:
 *   rho = relu(cos(s, F[a](:, r, c)))                     models/tracker.py:158-173
 *   (x, y) = TrackerHead(rho)                             models/networks/tracker_head.py:107-121
 * and out_xy[out_idx[m]] = (x, y) in PIXELS (model_inference.py:52 already applied), or normalised to [-1,1]
 * if normalized != 0 (the value Tracker.forward returns, tracker.py:303-325).
 * src_row / out_idx may be NULL (identity).  Any order of tgt is correct; sources sorted by tgt run fastest.
 * opts->method: DTK_TRACK_EXACT = fp32 everywhere (correlation volume staged through `workspace`), fully asynchronous;
 *               DTK_TRACK_MFMA  = fp16-MFMA correlation reduced on chip to peak records + fp32 window refinement.
 *                 Three tiers decide a source: (1) the no-fallback certificate, (2) whole-map
 *                 refiner statistics on the matrix cores for uncertified sources, (3) the exact path for sources the
 *                 fp16 pass cannot decide.  The sizes of tiers 2 and 3 are read back: this method SYNCHRONISES
 *                 `stream` once per call (twice if tier 2 ran; once more if `dM` is given).
 * opts->normalized: 0 = pixels, 1 = [-1,1].
 * opts->round_sources: sources per round of the MFMA pipeline (0 = default 4 194 304; rounded up to a multiple of 256).
 *                 The workspace size follows it.  Results do not depend on it (tests run several rounds with it).
 * opts->tier: DTK_TIER_AUTO, or DTK_TIER_WHOLE_MAP = skip the certificate, every source takes tier 2 (tests).
 * `dM` (device int32*, may be NULL): if given, the number of sources is min(M, *dM).
 * `stats` (HOST pointer, may be NULL): filled before returning; nothing is retained by the library between calls. */
#define DTK_TRACK_EXACT 0
#define DTK_TRACK_MFMA 1
#define DTK_TIER_AUTO 0
#define DTK_TIER_WHOLE_MAP 1
typedef struct dtk_track_opts {
    int32_t method, normalized, round_sources, tier;
    int32_t emb_rows;        /* number of rows of `emb` when sources go through src_row (0 = unknown): lets DTK_TRACK_MFMA convert
                              * the rows once per call instead of every round's sources */
} dtk_track_opts;
typedef struct dtk_track_stats {
    int32_t sources;         /* sources processed (min(M, *dM)) */
    int32_t whole_map_tier;  /* of which decided with whole-map statistics (tier 2) */
    int32_t exact_tier;      /* of which re-done on the exact fp32 path (tier 3) */
    int32_t syncs;           /* host synchronisations of `stream` this call made */
} dtk_track_stats;
size_t dtk_track_workspace_bytes(const dtk_geom* g, int M, const dtk_track_opts* opts);
int dtk_track(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16,
              const float* head, const float* emb, const int32_t* src_row, const int32_t* tgt,
              const int32_t* out_idx, float* out_xy, int M, const int32_t* dM, const dtk_track_opts* opts,
              dtk_track_stats* stats, void* workspace, size_t workspace_bytes, void* stream);

/* Arg-max cell of the RAW cosine map of each source against its target frame, and the cosine there: row arg-max of
 * the affinity matrix of the best-buddies preprocessing (preprocessing_dino_bb/extract_dino_best_buddies.py:36-39;
 * torch.argmax semantics: first maximum).  Sources are rows of `emb` like in dtk_track (src_row may be NULL); results go to
 * arg_cell[m], arg_cos[m].  DTK_TRACK_MFMA = fp16 candidate search + fp32 re-scoring, undecidable sources on the exact path
 * (one stream synchronisation); DTK_TRACK_EXACT = fp32 everywhere.  Workspace: dtk_track_workspace_bytes of the method. */
int dtk_argmax_cells(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16, const float* emb,
                     const int32_t* src_row, const int32_t* tgt, int32_t* arg_cell, float* arg_cos, int M, int method,
                     void* workspace, size_t workspace_bytes, void* stream);

/* DINO best-buddy ambiguity ratio (preprocessing_dino_bb/compute_dino_bb_nms.py:12-66): for each source m
 * (row src_row[m] of emb = the feature of a best-buddy cell) the cosine affinity row against frame tgt[m], its top-`topk`
 * entries (:14), boxes of +-box_size px around their cell centres (:21-27), torchvision batched_nms at iou_thresh (:30),
 * suppressed affinities zeroed (:34-36), then the two largest of the masked list -> peak_affs[m][0..1] (:38) and
 * r[m] = second / first (:42).  fp32 everywhere (exact-path correlation).  Workspace: dtk_bb_nms_workspace_bytes. */
size_t dtk_bb_nms_workspace_bytes(const dtk_geom* g, int M);
int dtk_bb_nms(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* src_row,
               const int32_t* tgt, float box_size, float iou_thresh, int topk, float* peak_affs, float* r, int M,
               void* workspace, size_t workspace_bytes, void* stream);

/* 16-bit copies of the feature volume consumed by DTK_TRACK_MFMA (C % 32 == 0), one buffer of dtk_feat_f16_bytes(g):
 *   - f16[t][row][col][c] = 32 F/|F|, every map row padded with zero cells to a multiple of 128 columns (an N-tile of the
 *     candidate GEMM is one map row);
 *   - at C = 384, behind it (256-byte aligned): the split planes of the window correlation, [t][cell][chunk of 32
 *     channels][hi 32 | lo 32] with s F = hi + lo (fp16 both) -- T*ph*pw*C*4 more bytes, made once per volume instead of
 *     once per staged element of every window box;
 *   - behind that (256-byte aligned), a 256-byte slot whose first float is s: 2^5, or the largest power of two with
 *     s * (largest cell norm of the volume) <= 2^14 when the features are large (round 6: DINOv2-like outlier statistics put
 *     components beyond 2047 = 65504 / 32) -- chosen on the device by dtk_make_feat_f16, read by the window-correlation kernels. */
size_t dtk_feat_f16_bytes(const dtk_geom* g);
int dtk_make_feat_f16(const dtk_geom* g, const float* feat, const float* norms, void* feat_f16, void* stream);

/* ---- K14: cosine similarity along trajectories -----------------------------------------------------------
 * ModelInference.compute_trajectory_cos_sims (models/model_inference.py:110-126):
 * cs[n][t] = cos(S[n][tq[n]], S[n][t]) with F.cosine_similarity semantics (eps 1e-8). S is [N][T][C]. */
int dtk_traj_cos_sims(const float* S, const int32_t* tq, float* cs, int N, int T, int C, void* stream);

/* ---- anchor bookkeeping (models/model_inference.py:156-165) ---------------------------------------------
 * From cs [N][T] and the anchor threshold build, on device and without host sync:
 *   n_anchors[N], pair_off[N+1] (pairs in n-major order, pair p of query n with its k-th anchor = pair_off[n]+k),
 *   and the source lists of the anchor stage sorted by anchor frame:
 *   src_row[m] = n*T + t, tgt[m] = a, out_idx[m] = p*T + t, for m < M_total = P*T (P = pair_off[N]).
 * counts[0] = P, counts[1] = M_total, counts[2] = number of queries with zero anchors (counts has 4 entries).
 * pair_frame must hold N*T entries, the three lists N*T*T entries (worst case), scratch 2*T+2 entries. */
int dtk_build_anchor_sources(const float* cs, float anchor_th, int N, int T, int32_t* n_anchors, int32_t* pair_off,
                             int32_t* pair_frame, int32_t* src_row, int32_t* tgt, int32_t* out_idx,
                             int32_t* counts, int32_t* scratch, void* stream);

/* ---- K15: occlusion (models/model_inference.py:169-200) ---------------------------------------------------
 * green [P][T][2] (anchor trajectories, pair-major), pair_off [N+1], pair_frame [P] (anchor frame of pair p),
 * traj [N][T][2], cs [N][T]  ->  occ [N][T] (uint8 0/1). */
int dtk_occlusion(const float* green, const int32_t* pair_off, const int32_t* pair_frame, const float* traj,
                  const float* cs, float anchor_th, float cos_th, uint8_t* occ, int N, int T, void* stream);

/* ---- TAP-Vid metric counts (eval/metrics.py:7-147 for one video) ----------------------------------
 * pred_tracks / gt_tracks [N][T][2] f32 (x, y), *_occluded [N][T] (uint8 0/1), query_frame [N]; coordinates are scaled
 * to the 256 x 256 evaluation raster by the four factors first (eval/metrics.py:204-211).  first_mode = 0: 'strided'
 * queries (every frame but the query frame is evaluated), 1: 'first'.  counts18 (device, uint64):
 * [0] evaluated, [1] occlusion agreements, [2] visible, then for thresholds 1, 2, 4, 8, 16 px:
 * [3+3i] within & visible, [4+3i] true positives, [5+3i] false positives.  The host turns them into
 * occlusion_accuracy, pts_within_*, jaccard_* and their averages (dino_tracker_amd/tapvid.py). */
int dtk_tapvid_counts(const float* pred_tracks, const uint8_t* pred_occluded, const float* gt_tracks,
                      const uint8_t* gt_occluded, const int32_t* query_frame, float pred_scale_x, float pred_scale_y,
                      float gt_scale_x, float gt_scale_y, int first_mode, int N, int T, unsigned long long* counts18,
                      void* stream);

/* ---- per-video test-time training: train-mode BatchNorm2d of the Delta-DINO CNN ------------------------
 * Replaces nn.BatchNorm2d in training mode inside DeltaDINO.forward (models/networks/delta_dino.py:38,53-55; batches of
 * <= 8 frames, models/tracker.py:118-124) and its autograd backward.  x, y, dy, dx: [N][C][HW] float32 (NCHW).
 * forward:  y = gamma (x - mean_c) rstd_c + beta over the batch statistics of channel c (biased variance, eps inside the
 *   root), followed by ReLU when relu != 0 (the nn.ReLU that follows the first three layers, delta_dino.py:41-42);
 *   save_mean / save_rstd [C] for the backward; running_mean / running_var [C] (nullable) are updated in place with
 *   `momentum` and the UNBIASED batch variance, as torch does.  Statistics by pairwise (Chan) merging of exact small-group
 *   moments -- no E[x^2] - E[x]^2.
 * pre_bias [C] (nullable): the bias of the convolution in front of the layer (delta_dino.py:29-31), NOT added to x by the
 *   caller: a per-channel constant only shifts the batch mean, so it enters running_mean and nothing else -- the layer
 *   never makes the (n, C, H, W) pass that adds it.
 * backward: dx, dgamma [C], dbeta [C] from dy (the gradient w.r.t. the ReLU'd output when relu != 0); dpre_bias [C]
 *   (nullable) = sum of dx = the gradient of pre_bias: zero in exact arithmetic, the float32 rounding residue here.
 * workspace: dtk_batchnorm_workspace_bytes(C) bytes of device memory. */
size_t dtk_batchnorm_workspace_bytes(int32_t C);
int dtk_batchnorm_train_forward(const float* x, const float* gamma, const float* beta, const float* pre_bias,
                                float* running_mean, float* running_var, float momentum, float eps, int32_t relu, float* y,
                                float* save_mean, float* save_rstd, int32_t N, int32_t C, int32_t HW, void* workspace,
                                size_t workspace_bytes, void* stream);
int dtk_batchnorm_train_backward(const float* x, const float* dy, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_rstd, int32_t relu, float* dx, float* dgamma,
                                 float* dbeta, float* dpre_bias, int32_t N, int32_t C, int32_t HW, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* BlurPool of the Delta-DINO CNN (antialiased_cnns.BlurPool(channels, stride=2): filt_size 4, reflect padding, depthwise
 * outer([1,3,3,1]) / 64; models/networks/delta_dino.py:43-44) and its adjoint.  x / dx: [planes][H][W], y / dy:
 * [planes][(H-1)/2+1][(W-1)/2+1], planes = N * C; H, W >= 4. */
int dtk_blurpool_forward(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream);
int dtk_blurpool_backward(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, void* stream);

/* ---- N1: the convolutions of the training step as fp32-grade matrix products on the fp16 matrix cores ---------------------
 * (models/networks/delta_dino.py:29-31 under autograd; dino_tracker.py:392-448).
 * dtk_gemm_nt_f32:  C[b][m][n] (+)= sum_k A[b][m][k] * B[b][n][k]     (both operands with the reduction index contiguous)
 *   fp32 in, fp32 out; every operand is split x = hi + lo (fp16) while it is staged and a product is hi.hi + hi.lo + lo.hi
 *   with fp32 accumulation (2^-22 relative per operand).  scale_a / scale_b: DEVICE scalars (powers of two, NULL = 1) applied
 *   before the split and divided out in the epilogue, so that small operands (gradients) keep normal fp16 halves.
 *   batch b: operands b * stride_a / stride_b (0 = shared), output b * stride_c.  split_k > 1 cuts the reduction into chunks
 *   whose partial sums are ATOMICALLY added to C; accumulate: 0 overwrite, 1 C += (single writer), 2 atomic add (several
 *   batches writing the same C: stride_c = 0).  With split_k > 1 or accumulate 2 the caller zeroes / owns C's prior content.
 * dtk_im2col: unfolded operand of a stride-1 'same' convolution (2 pad == dil (k - 1)), reflect or zero padding:
 *   layout 0: cols[f][l][Kp]  (l = y W + x, k = (c ksize + ky) ksize + kx contiguous, zero-filled to Kp)
 *   layout 1: cols[f][Kp][Lp] (pixels contiguous, zero-filled to Lp)
 * dtk_col2im: adjoint of layout 1 with Lp = H W (incl. the adjoint of the reflect padding): dcols[f][Kp][H W] -> dx[f][C][H][W]
 * dtk_transpose_f32: dst[b][c][r] = src[b][r][c]. */
int dtk_gemm_nt_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb,
                    int64_t ldc, int32_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t split_k,
                    int32_t accumulate, const float* scale_a, const float* scale_b, void* stream);
/* the same with an optional batch -> operand map: batch b reads B operand number index_b[b] and writes (accumulate 2: adds to)
 * output number index_c[b]; either may be NULL (identity). */
int dtk_gemm_nt_f32_indexed(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb,
                            int64_t ldc, int32_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t split_k,
                            int32_t accumulate, const float* scale_a, const float* scale_b, const int32_t* index_b,
                            const int32_t* index_c, void* stream);
int dtk_im2col(const float* x, float* cols, int32_t n, int32_t C, int32_t H, int32_t W, int32_t ksize, int32_t pad, int32_t dil,
               int32_t reflect, int32_t layout, int32_t Kp, int64_t Lp, void* stream);
int dtk_col2im(const float* dcols, float* dx, int32_t n, int32_t C, int32_t H, int32_t W, int32_t ksize, int32_t pad, int32_t dil,
               int32_t reflect, int32_t Kp, void* stream);
int dtk_transpose_f32(const float* src, float* dst, int64_t rows, int64_t cols, int32_t batch, void* stream);

/* CNN -> ViT grid alignment of the training step (models/utils.py:7-45), forward and backward, as a separable two-tap
 * resampling given by per-axis tables: destination index i reads source cells lo[i] and min(lo[i] + 1, n - 1) with weights
 * 1 - whi[i] and whi[i].  src / dsrc [planes][hs][ws], dst / ddst [planes][hd][wd].  Backward tables per axis: int32
 * ranges[4][n_src] = start, end of the destination indices whose LO is a, then start, end of those whose HI is a. */
int dtk_resample2d_forward(const float* src, float* dst, int64_t planes, int32_t hs, int32_t ws, int32_t hd, int32_t wd,
                           const int32_t* ylo, const float* ywhi, const int32_t* xlo, const float* xwhi, void* stream);
int dtk_resample2d_backward(const float* ddst, float* dsrc, int64_t planes, int32_t hs, int32_t ws, int32_t hd, int32_t wd,
                            const int32_t* yranges, const float* ywhi, const int32_t* xranges, const float* xwhi, void* stream);

/* Tracker.sample_embeddings of the training step (models/tracker.py:96-111 -> utils.py:65-109 for an exact frame index): pts [B][3] =
 * (x, y in [-1, 1] token-grid coordinates, frame index into the batch), feat the token-major batch embeddings [n][h w][C] ->
 * out [B][C], bilinear over the four surrounding cells, align_corners, border clamp.  The backward ACCUMULATES g [B][C] into dfeat
 * [n][h w][C] (atomic adds: several points share cells); the points carry no gradient (the reference detaches them, utils.py:91). */
int dtk_sample_bilinear_forward(const float* feat, const float* pts, float* out, int32_t B, int32_t n, int32_t h, int32_t w, int32_t C,
                                void* stream);
int dtk_sample_bilinear_backward(const float* g, const float* pts, float* dfeat, int32_t B, int32_t n, int32_t h, int32_t w, int32_t C,
                                 void* stream);

/* Operand scale of a gradient tensor for the fp16 split of the training convolutions: out[0] = 2^e with max |x| * 2^e in [2^9, 2^10]
 * (max |x| clamped below at 1e-30; a NaN in x gives NaN).  x: n floats on the device; scratch: 4 bytes on the device.  Three small
 * launches, no memset -- safe inside a captured graph (a library reduction is not, on this stack: csrc/common.h, dtk_zero_async). */
int dtk_pow2_scale(const float* x, int64_t n, float* out, void* scratch, void* stream);

/* ---- N1: the optimiser step of test-time training (dino_tracker.py:110-115: torch.optim.Adam over two parameter groups, default
 * betas / eps, no weight decay, no amsgrad; optimization/schedulers.py:4-8 scales the groups' learning rates) ---------------------
 * ONE launch updates every parameter tensor of the step: the tensors travel BY VALUE in the argument block (their gradients are
 * new allocations every iteration, so a device-side table would need a copy per step).  torch.optim.Adam's arithmetic:
 *   m <- m + (1 - beta1) (g - m);  v <- beta2 v + (1 - beta2) g g;
 *   p <- p - (lr[group] / (1 - beta1^step)) m / (sqrt(v) / sqrt(1 - beta2^step) + eps),   step = the tensor's own count. */
#define DTK_ADAM_MAX_TENSORS 32
#define DTK_ADAM_MAX_GROUPS 4
typedef struct dtk_adam_args {
    float* param[DTK_ADAM_MAX_TENSORS];            /* device, fp32, updated in place */
    const float* grad[DTK_ADAM_MAX_TENSORS];       /* device, fp32 */
    float* exp_avg[DTK_ADAM_MAX_TENSORS];          /* device, fp32, updated in place (m) */
    float* exp_avg_sq[DTK_ADAM_MAX_TENSORS];       /* device, fp32, updated in place (v) */
    int64_t numel[DTK_ADAM_MAX_TENSORS];
    int32_t group[DTK_ADAM_MAX_TENSORS];           /* index into lr[] */
    int32_t step[DTK_ADAM_MAX_TENSORS];            /* per tensor: 1-based step count AFTER this update (its bias corrections; a
                                                    * parameter that had no gradient in some iteration lags behind the others) */
    int32_t n_tensors;
    double lr[DTK_ADAM_MAX_GROUPS];                /* doubles, as torch keeps them: its scalar coefficients (1 - beta2 = 0.001, lr / (1 -
                                                    * beta1^step) ...) are formed in double and rounded to fp32 ONCE; 1.f - 0.999f is 4.7e-5 off */
    double beta1, beta2, eps;
} dtk_adam_args;
int dtk_adam_step(const dtk_adam_args* a, void* stream);
/* The same update for a CAPTURED iteration (hipGraph): pointers and sizes are baked into the launch, the two per-tensor scalars --
 * step size lr[group] / (1 - beta1^step) and 1 / sqrt(1 - beta2^step) -- are read from device memory that the host refreshes before
 * every replay.  dtk_adam_scalars forms them on the host exactly as dtk_adam_step does (out_host: 2 * DTK_ADAM_MAX_TENSORS floats,
 * step sizes first; it reads n_tensors, betas, eps, group[], step[] and lr[] only -- the tensor pointers may be null);
 * dtk_adam_step_dev launches with `scalars_dev` in that layout. */
int dtk_adam_scalars(const dtk_adam_args* a, float* out_host);
int dtk_adam_step_dev(const dtk_adam_args* a, const float* scalars_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DTK_H_ */
