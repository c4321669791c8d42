/*
 * fern.h -- C ABI of libfern.so: the MI355X (gfx950) implementation of FashionERN's
 * encode -> fuse -> rank inference path.
 *
 * The reference has no FFI: the path sits behind a Python object protocol
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it replaces
 * (paths relative to the reference repository root).  All pointers are raw device pointers
 * unless a parameter is called `host_*`; matrices are row-major, contiguous unless an ld* is
 * given; fp32 throughout (the reference evaluates in fp32: run/test/test_fiq.py:145,169).
 *
 * Conventions
 *   - every function returns 0 on success, a negative fern_status otherwise;
 *     fern_last_error() returns a message for the calling thread's last failure;
 *   - launch functions are asynchronous on `stream` (a hipStream_t passed as void*) and never
 *     synchronise; workspaces grow on first use (hipMalloc), so run one warm-up call per shape
 *     before capturing into a hipGraph;
 *   - a context is bound to one device and is not thread-safe (one per device per process);
 *   - re-finalising a weight group (fern_finalize_*) frees its previous device copy and invalidates forks made earlier;
 *   - NO collectives are exported.  SURVEY.md 8b listed fern_comm_init / fern_all_gather; they are deliberately left to
 *     torch.distributed (backend "nccl" = RCCL over xGMI, "gloo" in the CPU tests): the path has exactly one exchange
 *     step -- the all-gather of fused gallery shards -- and the reference side already owns a process group, so a second
 *     communicator inside this library would only duplicate its bootstrap.  The gallery-sharded alternative needs
 *     fern_sim_topk(idx_offset=) + fern_topk_merge from this library and two all-gathers from the caller.
 */
#ifndef FERN_H
#define FERN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FERN_ABI_VERSION 3

#if defined(__GNUC__)
#define FERN_API __attribute__((visibility("default")))
#else
#define FERN_API
#endif

typedef struct fern_ctx fern_ctx;

typedef enum {
    FERN_OK = 0,
    FERN_ERR_ARG = -1,      /* bad argument / unsupported shape */
    FERN_ERR_HIP = -2,      /* HIP runtime error */
    FERN_ERR_STATE = -3,    /* weights missing / not finalised */
    FERN_ERR_NOMEM = -4
} fern_status;

typedef enum { FERN_F32 = 0, FERN_I64 = 1 } fern_dtype;

/* which CombinerSimple / VisualSR instance of ERN (models/model.py:16-20, fusion_model.py:17-24) */
typedef enum {
    FERN_COMBINER_TARGET = 0,      /* ERN.Combiner_module      (model.py:20)  */
    FERN_COMBINER_DVR_GLOBAL = 1,  /* DVR.combiner_global      (fusion_model.py:22) */
    FERN_COMBINER_DVR_LOCAL = 2,   /* DVR.combiner_local       (fusion_model.py:23) */
    FERN_COMBINER_DVR_FINAL = 3    /* DVR.combiner             (fusion_model.py:24) */
} fern_combiner_id;

/* groups of fusion weights that can be finalised independently (a stand-alone CombinerSimple / VisualSR /
 * DVR_module loads its weights under the ERN prefix of the slot it occupies) */
typedef enum {
    FERN_PART_DVR = 1,              /* "DVR.*"              -> fern_dvr_fuse, DVR combiners / SR */
    FERN_PART_TARGET_SR = 2,        /* "SR_module.*"        -> fern_visual_sr(FERN_SR_TARGET) */
    FERN_PART_TARGET_COMBINER = 4,  /* "Combiner_module.*"  -> fern_combiner(FERN_COMBINER_TARGET) */
    FERN_PART_ALL = 7               /* everything ERN owns  -> + fern_index_fuse */
} fern_fusion_part;

/* Operand precision of the CLIP towers' token-level GEMMs (BASELINE.json config 5 names a reduced-precision encoder;
 * SURVEY.md 7 step 6 "perf mode").  FP32 is the parity mode (exact fp32 FMA chains on the fp32 MFMA) and the default.
 * BF16: in every full transformer block of both towers the QKV / out-proj / c_fc / c_proj contractions (and the last ViT
 * block's K/V projection) read activations and weights rounded to bf16 (RNE) and accumulate in fp32, and the attention
 * of those blocks runs in the bf16 operand form (fern_attention_bf16: q/k/v and the un-normalised softmax weights
 * rounded to bf16, fp32 accumulation); the two BERT blocks of the fusion stage (fern_dvr_fuse) follow the same recipe.
 * The residual streams, LayerNorm / softmax statistics, GELU, patch embedding, class-token chain of the last ViT block,
 * final projections, cross attention, VisualSR, the combiners and the ranking path stay fp32.  Rounding points are fixed by the layer structure, not by
 * the batch size, so results remain batch-invariant. */
typedef enum {
    FERN_PREC_FP32 = 0,
    FERN_PREC_BF16 = 1,
    /* BASELINE.json config 5 ("fp8 MFMA encoder path"): as BF16, but the four token-level GEMMs of a block (and the last ViT
     * block's K/V projection) take OCP e4m3fn operands on v_mfma_f32_32x32x16_fp8_fp8: one dynamic scale per token row
     * (max|row| / 448, computed where the row is produced) and one static scale per output channel of the weight, both
     * folded back in the fp32 epilogue; attention, and the fusion stage's BERT blocks, stay in the bf16 operand form (fp8 is
     * for the encoder GEMMs only).  Needs tower / MLP widths % 64 == 0. */
    FERN_PREC_FP8 = 2,
    /* The same block structure with BLOCK-SCALED (MX) fp8 operands on v_mfma_scale_f32_32x32x64_f8f6f4, gfx950's scaled MFMA at
     * twice the bf16 / plain-fp8 matrix rate: one E8M0 (power-of-two) scale per (token, 32 consecutive channels) of the
     * activations and per (output channel, 32 consecutive inputs) of the weights (fern_quantize_mx8), applied inside the MFMA --
     * an outlier channel costs the precision of its own 32-block, not of the whole token row.  The ViT patch embedding runs the same way
     * (patch rows quantised once, conv1 as a block-scaled GEMM) when 3 * patch^2 is a multiple of 128.  Needs tower / MLP widths % 128 == 0. */
    FERN_PREC_MX8 = 3,
    /* fp32 data everywhere (the FP32 mode's buffers, statistics, attention, epilogues), but every plain-epilogue GEMM of 256 rows or
     * more computes its fp32 dot products as "f32x3": both operands split in registers into three bf16 planes (8 + 8 + 8 mantissa bits,
     * by truncation) and the six partial products of total order <= 4 run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The
     * error against exact arithmetic is the fp32 MFMA kernel's own (~2e-6 of the output rms), at ~1.4-1.5x its speed; it is not the
     * sequential fp32 fma chain, so results are fp32-accurate but not bit-identical to FERN_PREC_FP32 (they are bit-identical across
     * tile shapes and batch sizes).  The ranking stage never uses it: fern_sim_topk* scores stay the exact chain.  No reference
     * counterpart (cuBLAS's own TF32x3-style modes are the closest relative). */
    FERN_PREC_F32X3 = 4,
    /* MX8 for the MLP pair (c_fc + c_proj: two thirds of a block's GEMM flops) of the IMAGE tower, BF16 for LayerNorm-1 / QKV /
     * attention / out-proj and (round 6) for the whole text tower, fp32 residual stream: the fp8 rounding points that remain sit where
     * the flops are (the ViT's 12608-row MLP GEMMs), none on the query's text side; between the two modes in speed and the fastest
     * mode that keeps Recall@50 within 1 pp of the fp32 encoder on bench.py's `reduced_modes` table -- the c5 default since round 6. */
    FERN_PREC_MX8_MLP = 5,
    /* Round 6, the c5 default: block-scaled fp8 (MX8) operands for ALL FOUR token-level GEMMs of the IMAGE tower -- LayerNorm-1 / -2 and
     * the attention kernel write e4m3fn + E8M0 scales, c_fc quantises its GELU output -- over the FP32 residual stream (not MX8's bf16
     * stream), and the bf16 block for the text tower and the fusion BERT.  All of the ViT's GEMM flops at the fp8 MFMA rate; on bench.py's
     * `reduced_modes` table (2 048 queries x 8 192 images) Recall@50 moves by 0.0 pp and the top-50 overlap is 0.942, against -2.7 pp /
     * 0.893 for FERN_PREC_MX8: what costs MX8 its Recall is the text tower's operands (-1.5 pp) and the bf16 residual stream (-1 pp),
     * not the image tower's fp8 GEMMs. */
    FERN_PREC_MX8_IMG = 6
} fern_precision;

typedef enum {
    FERN_SR_TARGET = 0,            /* ERN.SR_module            (model.py:19) */
    FERN_SR_DVR = 1                /* DVR.SR_module            (fusion_model.py:17) */
} fern_sr_id;

/* CLIP tower shapes (open_clip model config) */
typedef struct {
    int embed_dim;
    int image_size, patch_size, v_width, v_layers, v_heads, v_mlp;   /* ViT tower; v_layers == 0: none */
    int context_length, vocab_size, t_width, t_heads, t_layers, t_mlp;
    int v_arch;          /* 0 = ViT (fields above), 1 = open_clip ModifiedResNet (fields below; e.g. RN50x4) */
    int r_layers[4];     /* bottleneck blocks per stage, RN50x4: 4,6,10,6 */
    int r_width;         /* stem / stage-1 width, RN50x4: 80 */
    int r_heads;         /* attention-pool heads, RN50x4: 40 */
} fern_clip_config;

/* GEMM epilogues exposed for tests and for reference-side composition */
typedef enum {
    FERN_EPI_BIAS = 0,          /* C = A W^T + bias (bias may be NULL)            */
    FERN_EPI_BIAS_GELU = 1,     /* exact erf GELU                                 */
    FERN_EPI_BIAS_RELU = 2,
    FERN_EPI_BIAS_RESIDUAL = 3  /* C = A W^T + bias + R  (R has leading dim ldc)  */
} fern_epilogue;

/* kernel-time accounting for bench.py's roofline block */
typedef struct {
    double gemm_ms;        /* every fp32-MFMA GEMM launch: sum over its kernel dispatches of the dispatch's own begin -> end timestamps
                              (hipExtLaunchKernelGGL event pairs = what rocprofv3 --kernel-trace reports per kernel), not stream-marker intervals */
    double gemm_flops;     /* sum of 2*M*N*K of those launches */
    int64_t gemm_launches;
    double attn_ms;        /* attention launches, per-dispatch begin -> end timestamps like gemm_ms */
    double attn_flops;
    int64_t attn_launches;
    double topk_ms;        /* the rest of the ranking stage: sample pass, bound, candidate selection / rescoring, gated exact pass and every
                              launch boundary between them = (the stage's interval) - sweep_ms.  The interval is the stage's first dispatch
                              begin -> last dispatch end (dispatch timestamps, like gemm_ms) for fern_sim_topk_prefiltered and
                              fern_sim_topk_bf16, one stream-marker interval for the plain fern_sim_topk */
    int64_t topk_launches;
    double sweep_ms;       /* the full-gallery similarity sweep inside fern_sim_topk / _bf16 / _prefiltered: the kernel's own duration */
    double sweep_bytes;    /* algorithmic bytes of those sweeps: N*D*s_g + B*D*4 + B*K*8 */
    int64_t sweep_launches;
    double gemm_fp8_ms;    /* fp8-operand GEMM launches (FERN_PREC_FP8, fern_gemm_fp8): not included in gemm_* / gemm_bf16_* */
    double gemm_fp8_flops;
    int64_t gemm_fp8_launches;
    double gemm_bf16_ms;   /* bf16-operand GEMM launches (FERN_PREC_BF16 encoder blocks, fern_gemm_bf16): NOT included in gemm_* */
    double gemm_bf16_flops;
    int64_t gemm_bf16_launches;
    double gemm_alg_bytes; /* algorithmic bytes of the fp32 GEMM launches counted in gemm_*: 4 (M K + N K + M N) each */
    int64_t gemm_dispatches; /* kernel dispatches behind gemm_launches (a bulk + remainder plan is two dispatches per launch) */
    double gemm_mx8_ms;    /* block-scaled fp8 GEMM launches (FERN_PREC_MX8, fern_gemm_mx8): not included in any of the above */
    double gemm_mx8_flops;
    int64_t gemm_mx8_launches;
    double gemm_mx8_bf16_flops; /* the part of gemm_mx8_flops done on the BF16 MFMA: the text tower's GEMMs that ride in the image tower's
                                   block-scaled launches under FERN_PREC_MX8_IMG (fern_encode_pair) -- a roofline of those launches prices
                                   the two parts against their own peaks (ABI 3) */
} fern_prof_stats;

FERN_API int fern_abi_version(void);
FERN_API const char* fern_last_error(void);

/* lifetime ------------------------------------------------------------------------------ */
FERN_API int fern_ctx_create(int device, fern_ctx** out);
/* a second context on the same device that SHARES the parent's finalised weights (no copy) and owns its own workspace:
 * one per extra HIP stream when several batches are kept in flight.  Destroy forks before the parent; re-finalising the
 * parent invalidates them. */
FERN_API int fern_ctx_fork(fern_ctx* parent, fern_ctx** out);
FERN_API int fern_ctx_destroy(fern_ctx* ctx);
FERN_API int fern_sync(fern_ctx* ctx, void* stream);

/* weights: replaces nn.Module.load_state_dict (run/test/test_fiq.py:143,149) ----------------
 * `key` is the reference state-dict key (open_clip names for the CLIP
 * towers).  Data is copied; unknown keys are kept but unused (e.g. pooler, position_ids,
 * num_batches_tracked, logit_scale).  host_ptr is HOST memory. */
FERN_API int fern_load_tensor(fern_ctx* ctx, const char* key, const void* host_ptr, int dtype, int ndim,
                     const int64_t* shape);
/* repack for the kernels (packed QKV, folded BatchNorm, transposed projections) and upload */
FERN_API int fern_finalize_fusion(fern_ctx* ctx, int feature_dim, int parts); /* ERN(clip, feature_dim, device) model.py:8; parts = FERN_PART_* mask */
FERN_API int fern_finalize_clip(fern_ctx* ctx, const fern_clip_config* cfg); /* open_clip.create_model_and_transforms test_fiq.py:141 */

/* encoders ------------------------------------------------------------------------------ */
/* clip_model.encode_image(images) -- call site utils/utils.py:64, models/clip_model.py:13-15.
 * images [b,3,S,S] f32 NCHW -> out [b,embed_dim], un-normalised. */
FERN_API int fern_vit_encode_image(fern_ctx* ctx, const float* images, float* out, int b, void* stream);
/* clip_model.encode_text(text, mode=, visual_emb=) -- call sites run/test/test_fiq.py:102-103,
 * models/clip_model.py:23-31.  tokens [B,ctx] int64 -> out_global [B,D] (may be NULL) and
 * out_seq [B,ctx,D] (may be NULL); one tower pass serves both (SURVEY.md 8c definition).
 * visual_emb (may be NULL) is the reference's `visual_emb=ref_patch_feats.transpose(0, 1)` argument: a device pointer with
 * visual_emb_shape = HOST int64[3], which must equal {13, B, embed_dim} (FERN_ERR_ARG otherwise); its values are not read
 * (the text encoder that consumes them is unreleased, README.md:41 -- "vanilla CLIP single branch").
 * A token id outside [0, vocab_size) makes the caption's features NaN and is reported as FERN_ERR_ARG by fern_sync or by the
 * next fern_text_encode on the context (nn.Embedding raises in the reference; the launch path here never synchronises). */
FERN_API int fern_text_encode(fern_ctx* ctx, const int64_t* tokens, const float* visual_emb, const int64_t* visual_emb_shape,
                              float* out_global, float* out_seq, int B, void* stream);
/* Both towers of B composed queries in one pass (round 6): `clip_model.encode_image(images)` + `clip_model.encode_text(text)` of the same
 * batch -- the two calls every query batch of the reference's harness and of ERN makes (utils/utils.py:64, run/test/test_fiq.py:102-103,
 * models/clip_model.py:10-31) -- with the towers walked layer by layer so that the text layer's GEMMs ride in the image layer's launches
 * (models/others/modeling_clip.py:694-768 and :832-887 are independent until the fusion).  images [B,3,S,S], tokens [B,ctx] ->
 * out_image [B,D], out_global [B,D] (may be NULL), out_seq [B,ctx,D] (may be NULL).  Results are BIT-IDENTICAL to fern_vit_encode_image +
 * fern_text_encode.  With the ViT tower the launches are paired under FERN_PREC_FP32 (also under F32X3: two fp32-data-flow GEMMs per launch)
 * and under FERN_PREC_MX8_IMG (the image layer's block-scaled GEMM carries the text layer's bf16 GEMM); every other mode / tower simply
 * makes the two calls.  Token-id errors as fern_text_encode. */
FERN_API int fern_encode_pair(fern_ctx* ctx, const float* images, const int64_t* tokens, float* out_image, float* out_global, float* out_seq,
                              int B, void* stream);

/* fusion -------------------------------------------------------------------------------- */
/* ERN.forward(mode="test") = DVR_module.forward -- models/model.py:68-69, fusion_model.py:26-55 */
FERN_API int fern_dvr_fuse(fern_ctx* ctx, const float* ref_global /*[B,D]*/, const float* ref_local /*[B,13,D]*/,
                  const float* text_global /*[B,D]*/, const float* text_seq /*[B,77,D]*/,
                  float* out /*[B,D]*/, int B, int seq_len, void* stream);
/* ERN.forward(mode="index") -- models/model.py:64-66; normalize_input folds the caller's
 * F.normalize(index_features) (run/test/test_fiq.py:45).  Tiles over n internally. */
FERN_API int fern_index_fuse(fern_ctx* ctx, const float* tar_feats /*[n,D]*/, const float* tar_local /*[n,13,D]*/,
                    float* out /*[n,D]*/, int64_t n, int normalize_input, void* stream);
/* CombinerSimple.forward(image_features, text_features) -- fusion_model.py:86-94 */
FERN_API int fern_combiner(fern_ctx* ctx, int which, const float* image /*[n,D]*/, const float* text /*[n,D]*/,
                  float* out /*[n,D]*/, int64_t n, void* stream);
/* VisualSR.forward(local_feature) -- fusion_model.py:141-154 */
FERN_API int fern_visual_sr(fern_ctx* ctx, int which, const float* local /*[n,13,D]*/, float* out /*[n,D]*/,
                   int64_t n, void* stream);
/* Select the encoder precision for the calls that follow on this context (forks inherit the value at fork time).
 * No reference counterpart: the reference evaluates in fp32 only (run/test/test_fiq.py:141-149). */
FERN_API int fern_set_precision(fern_ctx* ctx, int precision /* fern_precision */);
FERN_API int fern_get_precision(fern_ctx* ctx);
/* models/others/Combiner_Model.py:6-70 (CLIP4Cir `Combiner`, not called by the reference's own scripts): weights are loaded
 * under the prefix "clip4cir." with that class's key names; image / text / out are [n, 2*clip_feature_dim]. */
FERN_API int fern_finalize_clip4cir(fern_ctx* ctx);
FERN_API int fern_combiner_clip4cir(fern_ctx* ctx, const float* image, const float* text, float* out, int64_t n, void* stream);
/* losses/loss.py:10-14 BatchBasedClassificationLoss.forward(predicted [B,D], target [B,D]) =
 * cross_entropy(100 * predicted @ target.T, labels = arange(B)), mean over the batch: forward VALUE only (the training
 * loop and its backward are outside this path; ERN's default mode, models/model.py:71-75, produces the two inputs).
 * out_loss: one float on the device.  D % 16 == 0. */
FERN_API int fern_batch_classification_loss(fern_ctx* ctx, const float* predicted, const float* target, int B, int D, float* out_loss,
                                   void* stream);
/* utils.element_wise_sum(image_features, text_features) = F.normalize(image + text) -- utils/utils.py:133-140 */
FERN_API int fern_element_wise_sum(fern_ctx* ctx, const float* image, const float* text, float* out, int64_t n, int d, void* stream);
/* F.normalize(x, dim=-1) -- run/test/test_fiq.py:45 */
FERN_API int fern_l2_normalize(fern_ctx* ctx, const float* x, float* out, int64_t n, int d, void* stream);

/* image side (data formats either side of the path) ----------------------------------------------
 * PIL `Image.resize` on 8-bit RGB, as called by utils/extract_fashioniq_patch.py:142-149 (`resize((360,360), ANTIALIAS)`)
 * and by torchvision `Resize(dim, BICUBIC)` in dataloader/dataset.py:73-87.  One separable pass each; `bounds` [out,2] =
 * (first tap, tap count) and `coeffs` [out,ksize] = fixed-point (2^-22) weights are DEVICE arrays computed by the host
 * exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do.  src is HWC u8 with `src_ld` pixels per row; (x0, y0)
 * is the origin of the region being resampled (`Image.crop`).  Bit-identical to PIL.
 * THREAD SAFETY (the one exception to "a context is not thread-safe"): these three entry points read only `ctx->device`, touch no
 * workspace, tuner or error state of the context beyond the calling thread's last-error slot, and launch on the stream they are
 * given -- dataset transforms may call them from loader worker threads (utils.ThreadedLoader) while another thread encodes. */
FERN_API int fern_resample_u8_horizontal(fern_ctx* ctx, const uint8_t* src, int64_t src_ld, int x0, int y0, int rows, uint8_t* dst /*[rows,ow,3]*/,
                                         int ow, const int32_t* bounds, const int32_t* coeffs, int ksize, void* stream);
FERN_API int fern_resample_u8_vertical(fern_ctx* ctx, const uint8_t* src, int64_t src_ld, int x0, int y0, int cols, uint8_t* dst /*[oh,cols,3]*/,
                                       int oh, const int32_t* bounds, const int32_t* coeffs, int ksize, void* stream);
/* torchvision ToTensor + Normalize (dataloader/dataset.py:84-86) with a crop window: n images [.,src_ld,3] u8, `src_image_stride`
 * bytes apart -> dst [n,3,oh,ow] f32 = (v/255 - mean[c]) / std[c]; mean / std are HOST arrays of 3 floats. */
FERN_API int fern_u8_to_normalized_chw(fern_ctx* ctx, const uint8_t* src, int64_t src_ld, int x0, int y0, int64_t src_image_stride, float* dst,
                                       int n, int oh, int ow, const float* host_mean, const float* host_std, void* stream);

/* rank ---------------------------------------------------------------------------------- */
/* distances = 1 - q @ g.T ; argsort(distances)[:, :K] -- run/test/test_fiq.py:49-50.
 * Scores are cosines (q.g), sorted descending, ties -> lower gallery index.  out_idx holds
 * local row + idx_offset; exclude_idx (may be NULL) [B] removes one global index per query
 * (CIRR reference removal, run/test/test_cirr.py:55-58).  Unfilled slots: score -inf, idx -1.
 * 1 <= K <= 64.  Exact for every gallery: the [B, N] scores are never stored -- a sampled bound filters the sweep into candidate
 * lists -- and a query whose lists overflow is ranked by a capacity-free exact pass inside the same call. */
FERN_API int fern_sim_topk(fern_ctx* ctx, const float* q /*[B,D]*/, const float* gallery /*[N,D]*/, int B,
                  int64_t N, int D, int K, float* out_scores /*[B,K]*/, int32_t* out_idx /*[B,K]*/,
                  int64_t idx_offset, const int32_t* exclude_idx, void* stream);
/* bf16-gallery variant (BASELINE.json config 5, "bf16 similarity"): `gallery` is [N,D] bf16 produced by fern_gallery_to_bf16
 * (round to nearest even); queries are rounded to bf16 inside the kernel; fp32 accumulation.  HBM-bound stream over the gallery.
 * Cosine scores differ from the fp32 path by <= ~5e-4 (inside north_star's 1e-3); ordering is exact for the rounded operands. */
FERN_API int fern_gallery_to_bf16(fern_ctx* ctx, const float* src, uint16_t* dst, int64_t n, int d, void* stream);
FERN_API int fern_sim_topk_bf16(fern_ctx* ctx, const float* q /*[B,D] f32*/, const uint16_t* gallery /*[N,D] bf16*/, int B, int64_t N, int D,
                                int K, float* out_scores, int32_t* out_idx, int64_t idx_offset, const int32_t* exclude_idx, void* stream);
/* The fp32-gallery ranking stage made HBM-bound: a CERTIFIED bf16 pre-filter + exact fp32 rescoring.  Same contract as fern_sim_topk --
 * `distances = 1 - q @ g.T; argsort` of run/test/test_fiq.py:49-50 on the fp32 gallery: the scores returned are the exact fp32 fma-chain
 * scores fern_sim_topk returns, the ordering is bit-identical to it -- but the pass over the gallery reads a bf16 copy (half the bytes,
 * bf16 MFMA rate), and only the few rows that can still be in the top-K are scored in fp32.
 *   fern_gallery_prepare: once per gallery (the reference builds its index once per evaluation, run/test/test_fiq.py:45-46):
 *       out_bf16 [N, D] = bf16(gallery) (round to nearest even = fern_gallery_to_bf16) and out_meta (DEVICE, 4 floats) =
 *       {max_n ||g_n - bf16(g_n)||, max_n ||bf16(g_n)||, max_n ||g_n||, 0}.  D % 4 == 0.  Re-run after the gallery changes.
 *   fern_sim_topk_prefiltered: per query b, |exact score - bf16 score| <= eps_b = ||q_b|| meta[0] + ||q_b - bf16(q_b)|| meta[1] + slack
 *       for EVERY row (Cauchy-Schwarz; slack = fp32 accumulation of both dot products), so every row of the exact top-K has a bf16
 *       score within 2 eps_b of the K-th best bf16 score: those rows -- K plus the few inside the margin -- are rescored with the exact
 *       chain from `gallery` and ranked.  Nothing about the result depends on the bf16 scores.  Queries whose candidates do not fit
 *       (galleries of near-ties) take fern_sim_topk's exact pass, as there.  D % 64 == 0 and D <= 768 use the pre-filter; other
 *       shapes run fern_sim_topk itself.  `gallery_bf16` / `meta` must come from fern_gallery_prepare on the same `gallery`. */
FERN_API int fern_gallery_prepare(fern_ctx* ctx, const float* gallery /*[N,D]*/, int64_t N, int D, uint16_t* out_bf16 /*[N,D]*/,
                                  float* out_meta /*[4], device*/, void* stream);
FERN_API int fern_sim_topk_prefiltered(fern_ctx* ctx, const float* q /*[B,D]*/, const float* gallery /*[N,D] f32*/,
                                       const uint16_t* gallery_bf16 /*[N,D]*/, const float* meta /*[4]*/, int B, int64_t N, int D, int K,
                                       float* out_scores /*[B,K]*/, int32_t* out_idx /*[B,K]*/, int64_t idx_offset,
                                       const int32_t* exclude_idx, void* stream);
/* Which form of fern_sim_topk_prefiltered's stage runs on this context (forks inherit at fork time).  Every form returns the same bits;
 * AUTO picks by a cost model of (B, N, D).  A tuning / test knob like fern_tuner_*: no reference counterpart. */
typedef enum {
    FERN_RANK_AUTO = 0,
    FERN_RANK_PLAIN = 1,   /* fern_sim_topk's fp32-MFMA sweep (the pre-filter copy is not read) */
    FERN_RANK_LISTS = 2,   /* bf16 sample pass -> bound - margin -> filtered bf16 sweep into candidate lists -> rescoring: large galleries */
    FERN_RANK_DENSE = 3    /* bf16 sweep storing its [B, N] scores -> one select + rescore kernel; offered while min(B, 1024) * N * 4 bytes of scores stay <= 1.1e9
                            * (~4.3M rows at B <= 64; up to 1.1 GB of workspace per lane) -- the cost model picks it up to ~600k rows at B = 64 */
} fern_rank_strategy;
FERN_API int fern_rank_set_strategy(fern_ctx* ctx, int strategy);
/* The bf16 sweep on its own: scores[b * ld + n] = the fp32-accumulated dot product of bf16(q[b]) and gallery_bf16[n] (v_mfma_f32_32x32x16_bf16,
 * k ascending) for every n < N -- the APPROXIMATE score the pre-filter selects on; tile_max (may be NULL) [B, ldt]: per 32 consecutive
 * gallery rows the largest of those scores (one wave tile of the sweep, reduced across its lanes), which is what the dense form of
 * fern_sim_topk_prefiltered reads instead of the rows.  ld >= N, ldt >= ceil(N / 32); D % 64 == 0, D <= 768.  For tests of the
 * certificate (|exact - approximate| <= eps_b) and for callers that want the whole approximate score matrix. */
FERN_API int fern_sweep_bf16_scores(fern_ctx* c, const float* q, const uint16_t* gallery_bf16, int B, int64_t N, int D, float* scores, int64_t ld,
                                    float* tile_max, int64_t ldt, void* stream);
/* scores of explicitly named gallery rows (CIRR subset ranking, run/test/test_cirr.py:64-66);
 * idx < 0 -> -inf */
FERN_API int fern_gather_scores(fern_ctx* ctx, const float* q /*[B,D]*/, const float* gallery /*[N,D]*/,
                       const int32_t* idx /*[B,m]*/, float* out /*[B,m]*/, int B, int m, int D,
                       void* stream);
/* merge R per-shard top-K lists (gallery sharded over R GPUs; SURVEY.md 8e alternative) */
FERN_API int fern_topk_merge(fern_ctx* ctx, const float* scores /*[R,B,K]*/, const int32_t* idx /*[R,B,K]*/,
                    float* out_scores /*[B,K]*/, int32_t* out_idx /*[B,K]*/, int R, int B, int K,
                    void* stream);

/* building blocks (exported for kernel-level parity tests) -------------------------------- */
/* C[M,N] = A[M,K] W[N,K]^T (+ epilogue); fp32 MFMA.  K % 32 == 0, lda/ldw % 4 == 0. */
FERN_API int fern_gemm(fern_ctx* ctx, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
              const float* residual, float* C, int64_t ldc, int M, int N, int K, int epilogue,
              void* stream);
/* bf16 operand form of fern_gemm ("perf mode" of the encoder GEMMs): A [M,lda] and W [N,ldw] hold bf16
 * bit patterns (fern_gallery_to_bf16 converts any fp32 buffer, round to nearest even), products accumulate in fp32 on
 * v_mfma_f32_32x32x16_bf16; C is fp32, or bf16 when out_bf16 != 0 (not with the residual epilogue).  K % 32 == 0,
 * lda/ldw % 8 == 0. */
FERN_API int fern_gemm_bf16(fern_ctx* ctx, const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias,
                   const float* residual, void* C, int64_t ldc, int M, int N, int K, int epilogue, int out_bf16,
                   void* stream);
/* fp8 (OCP e4m3fn) operand form: A [M,lda] / W [N,ldw] bytes with per-row scales (fern_quantize_rows_fp8 produces both:
 * scale[r] = max|row r| / 448, or 1 for a zero row; y = fp8(x / scale[r]), round to nearest even);
 * C = (sum_k A8 W8) * scale_a[row] * scale_w[col] + bias (+ GELU | + residual), fp32 accumulation on
 * v_mfma_f32_32x32x16_fp8_fp8.  K % 64 == 0, lda/ldw % 16 == 0; `x` of the quantiser is bf16 when x_is_bf16 else fp32,
 * d % 8 == 0, d <= 4096. */
FERN_API int fern_quantize_rows_fp8(fern_ctx* ctx, const void* x, int x_is_bf16, int64_t ldx, uint8_t* y, int64_t ldy, float* scale,
                           int64_t rows, int d, void* stream);
FERN_API int fern_gemm_fp8(fern_ctx* ctx, const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, int64_t ldw,
                  const float* scale_w, const float* bias, const float* residual, void* C, int64_t ldc, int M, int N, int K,
                  int epilogue, int out_bf16, void* stream);
/* MX (block-scaled) fp8 operand form, the operand format of gfx950's v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 / plain-fp8
 * matrix rate): OCP e4m3fn bytes with ONE E8M0 scale byte per (row, 32 consecutive k).  fern_quantize_mx8 produces both from a
 * bf16 or fp32 [rows, d] matrix: e = the smallest power-of-two exponent with max|block| * 2^-(e-127) <= 448 (clamped to [1, 253];
 * an all-zero block gets 1), y = fp8(x * 2^(127-e)) -- the scaling is exact, the cast rounds to nearest even.
 * Scale layout (uint8, (d / 128) * scale_rows * 4 bytes): byte of (row r, block b = k / 32) at ((b / 4) * scale_rows + r) * 4 + b % 4,
 * scale_rows >= rows.  fern_gemm_mx8: C = sum over blocks of 2^(ea-127) 2^(ew-127) (A8 . W8) + bias (+ GELU | + residual), fp32
 * accumulation, scales applied inside the MFMA.  K % 128 == 0, d % 128 == 0, d <= 4096, lda / ldw % 16 == 0, ldx / ldy % 8 == 0.
 * FERN_EPI_BIAS_RESIDUAL with out_bf16 != 0 is the bf16 RESIDUAL-STREAM form (FERN_PREC_MX8's token stream): `residual` then points
 * at bf16 [M, ldc] (may alias C) and C = bf16(sum + bias + float(residual)), one round-to-nearest-even per element. */
FERN_API int fern_quantize_mx8(fern_ctx* ctx, const void* x, int x_is_bf16, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* scales,
                      int64_t scale_rows, int64_t rows, int d, void* stream);
FERN_API int fern_gemm_mx8(fern_ctx* ctx, const uint8_t* A, int64_t lda, const uint8_t* scales_a, int64_t scale_rows_a, const uint8_t* W,
                  int64_t ldw, const uint8_t* scales_w, int64_t scale_rows_w, const float* bias, const float* residual, void* C,
                  int64_t ldc, int M, int N, int K, int epilogue, int out_bf16, void* stream);
/* fern_gemm_mx8 whose output is quantised where it is produced: C8 [M, ldc] e4m3fn bytes + scales_c (the layout above, scale_rows_c
 * rows) = fern_quantize_mx8 applied to the fp32 values bias + sum (+ GELU), bit for bit -- the A operand of the next
 * fern_gemm_mx8.  N % 128 == 0, ldc % 16 == 0; epilogue BIAS or BIAS_GELU. */
FERN_API int fern_gemm_mx8_quant(fern_ctx* ctx, const uint8_t* A, int64_t lda, const uint8_t* scales_a, int64_t scale_rows_a, const uint8_t* W,
                        int64_t ldw, const uint8_t* scales_w, int64_t scale_rows_w, const float* bias, uint8_t* C8, int64_t ldc,
                        uint8_t* scales_c, int64_t scale_rows_c, int M, int N, int K, int epilogue, void* stream);
/* y = LayerNorm(x (+ residual)) * gamma + beta, rows of width d */
FERN_API int fern_layernorm(fern_ctx* ctx, const float* x, const float* residual, const float* gamma,
                   const float* beta, float* y, int64_t rows, int d, float eps, void* stream);
/* softmax(scale * Q K^T (+causal)) V per (batch, head); q/k/v row strides in floats */
FERN_API int fern_attention(fern_ctx* ctx, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                   int64_t ldv, float* out, int64_t ldo, int batch, int heads, int head_dim, int s_q,
                   int s_k, int causal, float scale, void* stream);
/* bf16 operand form (perf mode of the CLIP towers): q/k/v/out hold bf16 bit patterns (strides in elements, % 8 == 0);
 * QK^T and PV on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, `scale` applied to the fp32 scores, softmax statistics
 * in fp32, the un-normalised weights rounded to bf16 for the PV product.  head_dim % 8 == 0, <= 96; s_k <= 224. */
FERN_API int fern_attention_bf16(fern_ctx* ctx, const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v,
                        int64_t ldv, uint16_t* out, int64_t ldo, int batch, int heads, int head_dim, int s_q, int s_k,
                        int causal, float scale, void* stream);

/* The GEMM launchers time their tile candidates once per new shape (all candidates are bit-identical).  This
 * returns the choices made so far in this process as text, one line per shape ("f32|bf16|fp8 M N K epilogue loader cfg");
 * the return value is the full length.  Setting FERN_GEMM_TILES=<file of such lines> before the first launch pins those
 * shapes (no timing runs, same kernels on every box).  No reference counterpart (cuBLAS picks its own kernels). */
FERN_API int64_t fern_tuner_export(char* buf, int64_t cap);
/* The reverse: `text` (NUL-terminated lines of the export format) replaces this process's choices for the shapes it lists, e.g.
 * rank 0's export broadcast to all ranks of a job so that every rank runs the same kernels (the step time of a multi-GPU job is
 * the max over ranks).  Lines that do not parse or name an inapplicable configuration are skipped.  Never changes a result. */
FERN_API int fern_tuner_import(const char* text);
/* The number of batches the caller keeps in flight on separate streams (the query pipeline's lanes; process-wide, default 1).
 * With more than one, the bf16 / fp8 / block-scaled GEMM families score a tile trial by duration x (share of the chip's
 * workgroup slots the launch fills)^0.75 instead of duration alone: a launch that leaves CUs to the other streams' kernels is worth more
 * to the pipeline than its own latency says.  Call before the first launch of a shape; tuned shapes keep their
 * choice.  The fp32 family is not affected.  Never changes a result.  No reference counterpart. */
FERN_API int fern_tuner_set_concurrency(int lanes);
/* Force ONE tile configuration of a GEMM family ("f32", "f32x3", "bf16", "fp8", "mx8"), process-wide, until cfg < 0 releases it (the
 * value of FERN_GEMM_CFG / _SPLIT_CFG / _BF16_CFG / _FP8_CFG / _MX8_CFG then applies again).  Every configuration of a family returns the
 * same bits, so this never changes a result: it exists so that one test process can walk all variants. */
FERN_API int fern_tuner_force_config(const char* family, int cfg);

/* Workspace generation of a context: incremented each time the context frees workspace memory it had handed to kernels before
 * (it consolidates its arena at the start of the next call after one that had to grow it).  A hipGraph captured from calls on
 * this context holds those addresses: re-capture it when the value differs from the one read right after the capture. */
FERN_API uint64_t fern_ws_generation(const fern_ctx* ctx);

/* profiling ----------------------------------------------------------------------------- */
/* While on, GEMM / attention / sweep launches are dispatched with a (start, stop) event pair each and timed by the dispatch's own begin /
 * end timestamps; the ranking STAGE (sample, bound, sweep, select, exact gate and the boundaries between them) is, for
 * fern_sim_topk_prefiltered / fern_sim_topk_bf16, the span first dispatch begin -> last dispatch end from those same timestamps, and for the
 * plain fern_sim_topk one stream-marker interval.  A launch that fails while instrumented leaves no record behind. */
FERN_API int fern_prof_enable(fern_ctx* ctx, int on);
FERN_API int fern_prof_collect(fern_ctx* ctx, fern_prof_stats* out);  /* synchronises, sums, resets */

#ifdef __cplusplus
}
#endif
#endif /* FERN_H */
