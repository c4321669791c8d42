// Internal launcher interface between the C-ABI layer (api.hip) and the kernel translation units.
// gfx950 only; no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#ifdef __HIPCC__
#include <hip/hip_ext.h>
#endif
#include <vector>

namespace fern {

// ---- kernel-precise launch timing (fern_prof_enable) ----------------------------------------------------------------------------
// While a LaunchTimer is armed on the calling thread (api.hip arms one around every GEMM / attention launch of an instrumented pass),
// FERN_LAUNCH dispatches through hipExtLaunchKernelGGL with a (start, stop) event pair: the events carry the DISPATCH's own begin /
// end timestamps -- what rocprofv3's kernel trace reports -- instead of the stream-marker interval of hipEventRecord, which added
// the marker packets and the dispatch latency (~3-5 us) to every launch (a fifth of a 25 us block-scaled GEMM).  Not armed: a plain
// hipLaunchKernelGGL, nothing else.
struct LaunchTimer {
    std::vector<hipEvent_t> events;      // start, stop, start, stop ... of the dispatches since the timer was armed
    std::vector<hipEvent_t>* pool;       // spare events (owned by the context)
};
extern thread_local LaunchTimer* g_launch_timer;
struct LaunchTimerPause {                 // the tile tuners' trial launches are not part of the launch being timed
    LaunchTimer* saved;
    LaunchTimerPause() : saved(g_launch_timer) { g_launch_timer = nullptr; }
    ~LaunchTimerPause() { g_launch_timer = saved; }
};
hipEvent_t launch_timer_event();         // an event from the armed timer's pool (created when the pool is empty)
#ifdef __HIPCC__
#define FERN_LAUNCH(kern, grid, block, shmem, stream, ...)                                                     \
    do {                                                                                                       \
        if (fern::g_launch_timer) {                                                                            \
            hipEvent_t ev_a_ = fern::launch_timer_event(), ev_b_ = fern::launch_timer_event();                 \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, ev_a_, ev_b_, 0, __VA_ARGS__);             \
            fern::g_launch_timer->events.push_back(ev_a_);                                                     \
            fern::g_launch_timer->events.push_back(ev_b_);                                                     \
        } else {                                                                                               \
            hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                                 \
        }                                                                                                      \
    } while (0)
#endif

// ---- fused similarity sweep + top-K selection (shared by the fp32 GEMM sweep and the bf16 sweep) ------------------
// The sweep never stores the [B, N] score matrix.  A first, small pass scores a jittered 1-in-R row SAMPLE of the gallery
// and takes each query's K-th best sample key as a lower bound of its true K-th best; the full sweep then appends only
// the scores that reach that bound (about K*R of N per query) to the query's candidate lists, and the final kernel
// selects the exact top-K from them.  Keys order by (score desc, gallery index asc) as one unsigned compare.
// A query owns RANK_SLOTS lists; gallery row n appends to list rank_slot(n) (n % RANK_SLOTS rotated by a hash of n / RANK_SLOTS).  So the append counters are spread over
// B * RANK_SLOTS addresses (returning atomics on ONE address serialise chip-wide at ~5 per us: with one counter per query the
// sweep was 15x slower than its HBM time), the lanes of a tile never collide, and a run of consecutive good rows -- near
// duplicates sit next to each other in real galleries -- lands in different lists instead of overflowing one.
constexpr int RANK_SLOTS = 256;
// list of gallery row n: inside every aligned block of 256 rows the mapping is a rotation (consecutive rows -- near duplicates sit
// next to each other in real galleries -- and the lanes of a tile land in different lists), and the rotation is a hash of the
// block number, so rows a multiple of 256 apart (a base set tiled up to 1M rows puts every copy of an item there) do NOT
// share a list.
#ifdef __HIPCC__
__host__ __device__
#endif
inline int rank_slot(long n) {
    const unsigned blk = (unsigned)(n >> 8);
    return (int)(((unsigned)n + ((blk * 0x9E3779B1u) >> 24)) & (RANK_SLOTS - 1));
}
struct TopkFilter {
    unsigned long long* cand;             // [B][RANK_SLOTS][cap] candidate keys, appended in arrival order
    const unsigned long long* thr_key;    // [B] lower bound: keys below it are not in the top-K (0: accept all, ~0: reject all)
    int* count;                           // [B][RANK_SLOTS] candidates offered so far (may exceed cap: overflow -> exact pass for that query)
    const int* exclude;                   // [B] gallery index to drop per query (CIRR reference removal), or null ...
    long exclude_off;                     // ... as a global index: the local row is exclude[q] - exclude_off
    int cap;                              // entries per list (<= 64: the select kernel reads a list with one wave load)
};
// sample column c -> gallery row: one row out of every run of R consecutive rows, at a hashed offset inside the run (a fixed
// stride would alias with periodic structure in the gallery order); monotonic in c, so sample order = gallery order
__host__ __device__ inline long sample_row(long c, int R) {
    if (R <= 1) return c;
    unsigned h = (unsigned)c * 2654435761u;
    h ^= h >> 15;
    return c * R + (long)(h % (unsigned)R);
}

// ---- GEMM: C[M,N] = A[M,K] * W[N,K]^T with fused epilogue (gemm.hip) ---------------------------
enum GemmEpi : int {
    EPI_BIAS = 0,            // C = acc + bias[col]                       (bias may be null)
    EPI_BIAS_GELU = 1,       // exact erf GELU
    EPI_BIAS_RELU = 2,
    EPI_BIAS_RESIDUAL = 3,   // C = acc + bias + R[row*ldc + col]
    EPI_COLAFFINE_TANH = 4,  // C = tanh((acc + bias[col]) * aux0[col] + aux1[col])   (Linear + BatchNorm1d(D) eval + Tanh)
    EPI_PATCH_EMBED = 5,     // C[(row + row/G2 + 1)] = acc + aux0[((row % G2) + 1)*N + col]   (conv1 patches + pos-emb, cls slot skipped)
    EPI_RELU_DOT = 6,        // partial[row][nb] = sum_col relu(acc + bias[col]) * aux0[col]       (Combiner hidden layer . w2)
    EPI_SR_LOCAL = 7,        // p = row % 13: v = tanh((acc + bias[col] - aux1[p]) * aux2[p] + aux3[p]);
                             // partial[row][nb] = sum_col v * G[(row/13)*ldg + col] * aux0[col]     (VisualSR local branch)
    EPI_BIAS_RESIDUAL_RELU = 8, // C = relu(acc + bias + R[row*ldc + col])   (ResNet bottleneck tail: conv3 + BN folded + identity)
    EPI_TOPK_FILTER = 9         // rows = queries, columns = gallery rows: nothing is stored; acc >= the query's bound is appended to p.filt
};
__host__ __device__ inline bool epi_is_reduce(int e) { return e == EPI_RELU_DOT || e == EPI_SR_LOCAL; }
// ALOAD_IM2COL: non-overlapping patches of an NCHW image (ViT conv1); ALOAD_CONV3: 3x3 / stride 1 / pad 1 window over an
// NHWC activation [B, conv_h, conv_w, conv_c] (k = (ky*3 + kx)*conv_c + c), out-of-image taps read from `zeros`.
enum GemmALoad : int { ALOAD_PLAIN = 0, ALOAD_IM2COL = 1, ALOAD_CONV3 = 2 };

struct GemmParams {
    const float* A;
    const float* W;
    float* C;
    const float* bias;
    const float* R;        // residual (EPI_BIAS_RESIDUAL)
    const float* aux0;
    const float* aux1;
    const float* aux2;
    const float* aux3;
    const float* G;        // EPI_SR_LOCAL: g_emb [n, ldg]
    float* partial;        // reduce epilogues: [M, nbn]
    long lda, ldw, ldc, ldg;
    int M, N, K;
    int epi, aload;
    int img, patch, grid;  // ALOAD_IM2COL: image side, patch side, patches per side; EPI_PATCH_EMBED uses grid*grid
    int conv_h, conv_w, conv_c;   // ALOAD_CONV3
    const float* zeros;           // ALOAD_CONV3: >= 64 bytes of zeros (16-byte aligned)
    // ksplit > 1: the k range is cut into ksplit equal slices (K % (ksplit * 64) == 0), blockIdx.y = slice; every slice stores its
    // raw accumulators (no bias / activation) to kpart[slice][M][N] and a reduce kernel adds the slices in ascending order and
    // applies the epilogue (launch_splitk_relu_dot).  The split is a property of the CALL SITE (a function of N and K only,
    // never of M), so a row's bits do not depend on the batch it travels in; every tile configuration walks a slice in the
    // same k order, so the tuner stays free.  Plain loader only.
    int ksplit;
    float* kpart;
    int w_sample;                 // > 1: W row r is gallery row sample_row(r, w_sample) (the sample pass of the fused top-K sweep)
    TopkFilter filt;              // EPI_TOPK_FILTER
    const int* gate;              // when set: the launch does nothing unless *gate != 0 (retry pass of the fused top-K sweep)
    // split == 3: "f32x3" arithmetic (FERN_PREC_F32X3): fp32 operands split into three bf16 planes in registers, six bf16 MFMAs per
    // pair of fp32 ones -- fp32-accurate (error vs exact arithmetic = the fp32 kernel's), not the fp32 fma chain.  Plain loader,
    // stored (non-reduce) epilogues, no split-K, M >= 256; anything else runs the fp32 kernels.
    int split;
    // bf16 operand form (launch_gemm_bf16): A [M, lda] and W [N, ldw] hold bf16 bit patterns, strides in elements
    const unsigned short* Ab;
    const unsigned short* Wb;
    int out_bf16;                 // plain epilogues: store C as bf16 (ldc in elements) instead of fp32
    // block-scaled family, EPI_BIAS_RESIDUAL with out_bf16: the residual stream itself is bf16 (Rb [M, ldc] bf16 in, C bf16 out,
    // may alias): C = bf16(acc + bias + float(Rb)) -- one rounding per residual add (FERN_PREC_MX8's token stream, api.hip)
    const unsigned short* Rb;
    // fp8 (OCP e4m3fn) operand form: Ab / Wb point at fp8 bytes (strides in elements = bytes), K % 64 == 0; the quantisation
    // scales are folded back in the epilogue: C = acc * scale_a[row] * scale_w[col] (+ bias ...)
    int fp8;
    const float* scale_a;         // [M] per-row (token) activation scales, or null
    const float* scale_w;         // [N] per-output-channel weight scales, or null
    // fp8 == 2: block-scaled (MX) form -- one E8M0 scale byte per (row, 32 consecutive k) instead of scale_a / scale_w, applied
    // inside v_mfma_scale_f32_32x32x64_f8f6f4.  mxa / mxw use the tile layout mx_scale_offset() below; K % 128 == 0.
    const unsigned char* mxa;     // A's scales, mxa_rows >= M rows per 128-k tile
    const unsigned char* mxw;     // W's scales, mxw_rows >= N
    long mxa_rows, mxw_rows;
    // out_mx8 (fp8 == 2, EPI_BIAS / EPI_BIAS_GELU, N % 32 == 0): the output is quantised where it is produced -- C holds e4m3fn
    // bytes (ldc in bytes), mxc the E8M0 scales of its 32-column blocks (mx_scale_offset layout, mxc_rows rows): the next
    // GEMM's A operand, with the same rounding as launch_quantize_mx8 applied to the fp32 values.
    int out_mx8;
    unsigned char* mxc;
    long mxc_rows;
#ifdef FERN_GEMM_TRACE
    // tools/probe/gemm_timeline.hip only (the library is never built with this macro): per-wave cycle stamps of the LDS-DMA
    // GEMM kernel -- [workgroup][wave][FERN_GEMM_TRACE_SLOTS] = {hw id, xcc id, realtime at entry, realtime at exit, cycle
    // counter at entry, after the prologue barrier, after the barrier of every k tile, after the epilogue}
    long long* trace;
    // layout experiment of the probe: operands stored k-slab-major, [K/16][rows][16] (a tile's 16-k slice of 128 rows is 8 KiB
    // contiguous: every LDS-DMA instruction reads whole 128-byte lines) instead of row-major [rows][K]
    int packed;
#endif
};
#ifdef FERN_GEMM_TRACE
constexpr int FERN_GEMM_TRACE_SLOTS = 256;
#endif
// MX scale layout of a [rows, K] fp8 matrix: scale byte of (row r, 32-k block b) lives at
//   ((b / 4) * srows + r) * 4 + (b % 4)      (srows >= rows: the array's row count)
// i.e. one dword per (128-k tile, row) -- what one GEMM workgroup stages per k tile is contiguous over its rows.
#ifdef __HIPCC__
__host__ __device__
#endif
inline long mx_scale_offset(long r, long b, long srows) { return ((b >> 2) * srows + r) * 4 + (b & 3); }
// Number of column blocks the reduce epilogues write per row (depends on the tile chosen for this shape).
int gemm_num_col_blocks(int M, int N, int K);
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
hipError_t launch_gemm_pair(const GemmParams& p1, const GemmParams& p2, hipStream_t s);      // two plain GEMMs, ONE launch when p1's tuned plan is a mixed plan (gemm.hip)
int gemm_last_dispatches();      // kernel dispatches of the calling thread's last launch_gemm (2 for a bulk + remainder plan)
// bf16 x bf16 -> fp32-accumulate GEMM (v_mfma_f32_32x32x16_bf16); plain epilogues only, K % 32 == 0, ALOAD_PLAIN
hipError_t launch_gemm_bf16(const GemmParams& p, hipStream_t s);
hipError_t launch_gemm_mxbf_pair(const GemmParams& p_mx8, const GemmParams& p_bf16, hipStream_t s);      // image (block-scaled) + text (bf16) GEMM of a layer, ONE launch where it wins
int gemm_bf16_last_dispatches();
hipError_t launch_gemm_pp(bool mx, const GemmParams& p, hipStream_t s);      // gemm_pp.hip: the ping-pong 256 x 256 tile (cfg 7 / 11 of the bf16 / block-scaled families)
// the per-shape tile choices made so far, one text line per shape (the format FERN_GEMM_TILES=<file> reads back)
void gemm_tuner_export(std::string& out);
void gemm_bf16_tuner_export(std::string& out);
// the reverse: lines of that format replace this process's choices for the listed shapes
void gemm_tuner_import(const std::string& text);
void gemm_bf16_tuner_import(const std::string& text);
// reduced-precision families: score trials for a pipeline that keeps `n` batches in flight on separate streams (gemm_bf16.hip)
void gemm_bf16_tuner_set_concurrency(int n);
// Force one tile configuration of a family at run time (family 0 fp32, 1 f32x3: gemm.hip; 2 bf16, 3 fp8, 4 block-scaled fp8: gemm_bf16.hip);
// cfg < 0 returns to the environment's value (FERN_GEMM_CFG, ...).  Every configuration of a family gives the same bits.
bool gemm_force_cfg(int family, int cfg);
bool gemm_bf16_force_cfg(int family, int cfg);

#ifdef __HIPCC__
// 64-bit ranking keys: orderable(score) << 32 | ~index, so "score descending, index ascending" is one unsigned compare
__device__ __forceinline__ unsigned orderable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) {
    return ((unsigned long long)orderable(score) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}
// the score part of a bound as the float the fast reject compares with: !(acc < bound) lets every key >= thr_key through
// (and NaNs, which rank first as keys); 0 = accept all -> NaN, ~0 = reject all -> +inf
__device__ __forceinline__ float filter_bound(unsigned long long thr_key) {
    return thr_key == ~0ull ? __builtin_inff() : unorderable((unsigned)(thr_key >> 32));
}
// One 32x32 accumulator tile of scores (register r = query q0 + (r & 3) + 8 (r >> 2) + 4 lh, lane = gallery row n): the
// common case -- no score reaches its query's bound -- is 16 compares and one branch.  Survivors: the exact key test, then
// all the tile's list appends are issued back to back (one round trip for the returned positions, not one per register).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void topk_filter_tile(const f32x16_t& acc, const float (&bound)[16], int q0, int lh, long n, bool n_ok, int B,
                                                 const TopkFilter& f) {
    unsigned hits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) hits |= (!(acc[r] < bound[r]) ? 1u : 0u) << r;
    hits = n_ok ? hits : 0u;
    if (!__any(hits != 0)) return;
    const long slot = rank_slot(n);
    const bool use_ex = f.exclude != nullptr;
    // eight registers at a time: their appends are issued back to back (one round trip for the returned positions), and the
    // live state stays small -- with all 16 positions live the 128-VGPR GEMM kernels spilled
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {
        if (!__any((hits >> r0) & 0xFFu)) continue;
        int pos[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            pos[u] = -1;
            const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if ((hits >> r & 1u) && q < B) {
                const unsigned long long key = make_key(acc[r], (unsigned)n);
                if (key >= f.thr_key[q] && !(use_ex && (long)f.exclude[q] - f.exclude_off == n))
                    pos[u] = atomicAdd(&f.count[(long)q * RANK_SLOTS + slot], 1);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (pos[u] >= 0 && pos[u] < f.cap) f.cand[((long)q * RANK_SLOTS + slot) * f.cap + pos[u]] = make_key(acc[r], (unsigned)n);
        }
    }
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// four floats -> four OCP e4m3fn bytes (v_cvt_pk_fp8_f32: round to nearest even; inputs are pre-scaled into [-448, 448])
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}
// MX quantiser arithmetic (elem.hip quantisers, gemm_bf16.hip quantising epilogue): the E8M0 byte of a block with maximum `amax`,
// and the exact power-of-two factor that scales the block into e4m3 range
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    const unsigned u = __float_as_uint(amax);
    const int e = (int)(u >> 23) - 8 + ((u & 0x7FFFFFu) > 0x600000u ? 1 : 0);
    return (unsigned)min(max(e, 1), 253);
}
__device__ __forceinline__ float mx_inv_scale(unsigned e) { return __uint_as_float((254u - e) << 23); }
// fp32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 (one instruction per PAIR of values on gfx950; the integer form
// u += 0x7fff + ((u >> 16) & 1) it replaces is three per value and gives the same bits for every finite input)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned f32x2_to_bf16x2_bits(float lo, float hi) {      // lo in bits 0-15
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
#endif

// ---- attention (attn.hip) ---------------------------------------------------------------------
struct AttnParams {
    const float* q; const float* k; const float* v; float* out;
    long ldq, ldk, ldv, ldo;     // row strides (floats)
    int batch, heads, hd, s_q, s_k, causal;
    float scale;
    unsigned short* out_b;       // when set: the output is stored as bf16 here (ldo in elements) instead of fp32 `out`
    // bf16 operand form (all three set, with out_b): q/k/v hold bf16 bit patterns, strides in elements; QK^T and PV run on
    // v_mfma_f32_32x32x16_bf16 with fp32 accumulation, softmax statistics in fp32, P rounded to bf16 for the PV product
    const unsigned short* qb;
    const unsigned short* kb;
    const unsigned short* vb;
    // bf16 operand form with a block-scaled fp8 output instead of out_b (MX operand of the out-projection; hd % 32 == 0): e4m3fn
    // bytes (ldo in bytes) + E8M0 scales in the mx_scale_offset layout, quantised from the fp32 output values
    unsigned char* out_q8;
    unsigned char* out_scales;
    long out_srows;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);   // hipErrorInvalidValue for unsupported shapes

// ---- row-wise / element-wise kernels (elem.hip) -------------------------------------------------
hipError_t launch_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                            long rows, int d, long ldx, long ldy, float eps, hipStream_t s);
// same statistics, output rounded to bf16 (operand of the bf16 encoder GEMMs)
hipError_t launch_layernorm_bf16(const float* x, const float* gamma, const float* beta, unsigned short* y, long rows, int d, long ldx,
                                 long ldy, float eps, hipStream_t s);
// fp8 (e4m3fn) quantisation with one scale per row: scale[r] = max|row| / 448 (1 for an all-zero row), y = fp8(x / scale[r]).
// LayerNorm fused form, bf16-row form (attention / GELU outputs), fp32-row form (weights, [N, K]: one scale per output channel).
hipError_t launch_layernorm_fp8(const float* x, const float* gamma, const float* beta, unsigned char* y, float* scale, long rows, int d,
                                long ldx, long ldy, float eps, hipStream_t s);
hipError_t launch_quantize_rows_fp8(const unsigned short* x_bf16, const float* x_f32, long ldx, unsigned char* y, long ldy, float* scale,
                                    long rows, int d, hipStream_t s);
// MX (block-scaled) e4m3fn quantisation: per (row, 32 consecutive k) one E8M0 byte e = smallest power of two with
// max|block| * 2^-(e-127) <= 448 (clamped to [1, 253]; an all-zero block gets 1), y = fp8(x * 2^(127-e)) (exact scaling, RNE cast).
// Scales in the mx_scale_offset layout with `srows` rows.  d % 128 == 0, d <= 4096.  LayerNorm-fused, bf16-row and fp32-row forms.
hipError_t launch_layernorm_mx8(const float* x, const float* gamma, const float* beta, unsigned char* y, unsigned char* scales, long srows,
                                long rows, int d, long ldx, long ldy, float eps, hipStream_t s, const unsigned short* x_bf16 = nullptr /* when set: the rows are bf16 (x unused) */);
hipError_t launch_quantize_mx8(const unsigned short* x_bf16, const float* x_f32, long ldx, unsigned char* y, long ldy, unsigned char* scales,
                               long srows, long rows, int d, hipStream_t s);
// patch rows [b * grid * grid, 3 * patch * patch] of an image batch ((channel, y, x) order = conv1's weight layout), MX-quantised
hipError_t launch_im2col_bf16(const float* images, unsigned short* y, int b, int img, int patch, int grid, hipStream_t s);
hipError_t launch_im2col_mx8(const float* images, unsigned char* y, unsigned char* scales, long srows, int b, int img, int patch, int grid,
                             hipStream_t s);
// mode 0: x / max(||x||, eps) (F.normalize); mode 1: x / (||x|| + eps) (VisualSR.l2norm)
hipError_t launch_l2norm(const float* x, long ldx, float* y, long ldy, long rows, int d, float eps, int mode, hipStream_t s,
                         const float* x2 = nullptr);   // x2: optional addend (normalize(x + x2))
// y[i, :] = mean_{p<P} x[i*group_stride + row_add + p, :]
hipError_t launch_mean_rows(const float* x, long ldx, float* y, long ldy, long n, int P, int d, long group_stride, long row_add,
                            hipStream_t s);
// y[i, :] = x[(i / group)*group_stride + (i % group) + (idx ? idx[i] : row_add), :]
// out[0] = mean_r( logsumexp_c(scale * logits[r, c]) - scale * logits[r, r] ), r, c < n; row_loss: [n] scratch
hipError_t launch_ce_diag_mean(const float* logits, long ld, int n, float scale, float* row_loss, float* out, hipStream_t s);
hipError_t launch_gather_rows(const float* x, long ldx, float* y, long ldy, long n, int d, int group, long group_stride, long row_add,
                              const int* idx, hipStream_t s);
// y[i] (fp32, ldy) = the bf16 row i * row_stride of x (ldx elements per row), widened exactly
hipError_t launch_gather_rows_bf16(const unsigned short* x, long ldx, float* y, long ldy, long n, int d, long row_stride, hipStream_t s);
// BERT embeddings of the fusion encoder: X[b,s] = LN(cat(cls, local, seq)[b,s] + type[s >= P+1] + pos[s]) (eps 1e-12)
hipError_t launch_bert_embed(const float* cls, const float* local, const float* seq, const float* type_emb,
                             const float* pos_emb, const float* gamma, const float* beta, float* X,
                             int B, int P, int T, int d, float eps, hipStream_t s);
// CLIP text embeddings: X[b,s] = tok_emb[text[b,s]] + pos[s]; also eot[b] = argmax_s text[b,s].  A token id outside
// [0, vocab) poisons its row with NaN and stores 1 + its flat position in *bad_flag (host-mapped, may be null).
hipError_t launch_text_embed(const int64_t* text, const float* tok_emb, const float* pos_emb, float* X, int* eot,
                             int B, int T, int d, int vocab, int* bad_flag, hipStream_t s);
// ViT class rows: X[b, 0, :] = cls + pos[0]
hipError_t launch_vit_cls(const float* cls, const float* pos, float* X, int B, int tokens, int d, hipStream_t s);
// CombinerSimple tail: s = sigmoid(sum_nb partial[row][nb] + b2); out = normalize(s*text + (1-s)*image)
hipError_t launch_combiner_finalize(const float* partial, int nb, const float* b2, const float* image, const float* text,
                                    float* out, long n, int d, hipStream_t s, const float* extra = nullptr);
// VisualSR tail: logits[p] = sum_nb partial[row*13+p][nb] + bc; w = softmax_13; new = sum w_p local_p; out = new/(||new||+1e-8)
hipError_t launch_sr_finalize(const float* partial, int nb, const float* bc, const float* local, float* out,
                              long n, int d, hipStream_t s);
// ModifiedResNet stem conv1: 3x3 / stride 2 / pad 1 from NCHW [B,3,S,S] to NHWC [B,S/2,S/2,cout_pad], folded BN + ReLU
hipError_t launch_stem_conv(const float* img, const float* w /*[cout_pad,27] (c,ky,kx)*/, const float* bias, float* out, int B, int S,
                            int cout_pad, hipStream_t s);
// NHWC average pool k x k, stride k: [B,H,W,C] -> [B,H/k,W/k,C]
hipError_t launch_avgpool_nhwc(const float* x, float* y, int B, int H, int W, int C, int k, hipStream_t s);
// AttentionPool2d tokens: mean[b] = mean_t x[b,t]; T[b,0] = mean[b] + pos[0]; T[b,1+t] = x[b,t] + pos[1+t]; T0[b] = T[b,0]
hipError_t launch_attnpool_tokens(const float* x, float* mean, const float* pos, float* T, float* T0, int B, int HW, int C, hipStream_t s);
// scores[b, j] = q[b] . gallery[idx[b, j]]  (idx < 0 -> -inf)
hipError_t launch_gather_scores(const float* q, const float* gallery, const int* idx, float* out, int B, int m, int d,
                                hipStream_t s);

// split-K tail of the CombinerSimple hidden layer: partial[row][g] = sum over the 32 columns of group g of
// relu(sum_s kpart[s][row][col] + bias[col]) * w2[col]  -- the layout launch_combiner_finalize reads (N % 32 == 0)
hipError_t launch_splitk_relu_dot(const float* kpart, int S, long M, int N, const float* bias, const float* w2, float* partial, hipStream_t s);
// ... and of a bias (+ residual) GEMM: C = (slices of kpart [S][M][N], added in ascending order) + bias + R (R may be null / alias C)
hipError_t launch_splitk_bias_residual(const float* kpart, int S, long M, int N, const float* bias, const float* R, long ldr, float* C, long ldc,
                                       hipStream_t s);

// ---- 8-bit image resampling / tensor conversion (image.hip) -------------------------------------------------------
hipError_t launch_resample_h(const unsigned char* src, long src_ld, int x0, int y0, int rows, unsigned char* dst, int ow, const int* bounds,
                             const int* kk, int ksize, hipStream_t s);
hipError_t launch_resample_v(const unsigned char* src, long src_ld, int x0, int y0, int cols, unsigned char* dst, int oh, const int* bounds,
                             const int* kk, int ksize, hipStream_t s);
hipError_t launch_u8_to_chw(const unsigned char* src, long src_ld, int x0, int y0, float* dst, int n, long src_img_stride, int oh, int ow,
                            const float* mean, const float* stdv, hipStream_t s);

// ---- bf16 gallery sweep (sweep_bf16.hip) ---------------------------------------------------------------------------
hipError_t launch_f32_to_bf16(const float* x, unsigned short* y, long n, hipStream_t s);
hipError_t launch_bf16_to_f32(const unsigned short* x, float* y, long n, hipStream_t s);      // exact (n % 4 == 0)
// y = bf16(x) (RNE) and meta[0..2] = max over rows of ||x_n - y_n||, ||y_n||, ||x_n|| (meta[3] = 0): what certifies y as a pre-filter
hipError_t launch_gallery_prepare(const float* x, unsigned short* y, long n, int d, float* meta, hipStream_t s);
// q < B <= 64 queries against a bf16 gallery [N, D] (D % 64 == 0), fp32 accumulation.
// Sample form (filt == null): scores[q, c] = Q[q] . G[sample_row(c, R)] for c < S, row stride ld.
// Filter form (filt != null): nothing is stored, survivors go to filt (gate as GemmParams.gate).
// zero_flags (sample form only, may be null): two ints the first workgroup zeroes (the ranking stage's overflow flags, when no bound
// kernel runs before the selection).
hipError_t launch_sweep_bf16(const float* q, const unsigned short* g, float* scores, long ld, int B, long N, int D, long S, int R,
                             const TopkFilter* filt, const int* gate, hipStream_t s, int* zero_flags = nullptr, float* tmax = nullptr, long ldt = 0);

// ---- top-K (topk.hip) ------------------------------------------------------------------------
// Fused sweep, step 2: per query the K-th best key of the sample scores [B, ld] (S valid columns; column c is gallery row
// sample_row(c, R)) -> thr_key[b] (0 when the sample holds fewer than K rows); also resets count[b][*] and flags[0..1].  A sample
// row that is the query's excluded gallery index (exclude[b] - exclude_off, exclude may be null) does not count.
// Pre-filter form (q != null; fern_sim_topk_prefiltered): the sample scores are bf16-sweep approximations of the exact fp32 scores; the
// published bound is lowered by the certified margin 2 eps_b (topk.hip: BoundMargin; meta = launch_gallery_prepare's {E, G~, G}) and
// margin_out[b] keeps the margin for launch_topk_rescore.
hipError_t launch_topk_sample_bound(const float* scores, long ld, int B, long S, int R, int K, const int* exclude, long exclude_off,
                                    unsigned long long* thr_key, int* count, int* flags, int* state, hipStream_t s, const float* q = nullptr,
                                    int D = 0, const float* meta = nullptr, float* margin_out = nullptr);
// Final step of the certified pre-filter: T~ = K-th best approximate key of each query's lists, survivors = rows within `margin` of it,
// exact fp32 fma-chain score (the sweep's k order) of every survivor from the fp32 gallery, exact top-K of those.  Queries without
// room (list overflow, > 6144 candidates, > 1024 survivors) are flagged for launch_rank_exact.  D % 32 == 0, D <= 1024.
// Dense form for small galleries: `approx` [B, ld] holds the bf16 sweep's score of EVERY row (launch_sweep_bf16 in its sample form with
// S = N, R = 1); one kernel does bound, collection, T~, survivors, exact rescoring and ranking per query.  Sets state[b] / done[b] and
// (when a query has no room) thr_key[b] + flags[0] for launch_rank_exact; flags[0..1] must be zero before (launch_sweep_bf16's
// zero_flags does it).
// inline_exact != 0: a query without room is ranked exactly by its own workgroup (small galleries: no gated exact-pass launch needed).
// tmax != null ([B, ldt]: launch_sweep_bf16's tile maxima, one per 32 gallery rows; ld must cover whole tiles): galleries of >= 16 384
// rows select on the tile maxima and read only the listed tiles' scores (topk_tiles_rescore_kernel) -- same results.
hipError_t launch_topk_dense_rescore(const float* approx, long ld, long N, const float* q, const float* gallery, int D, const float* meta,
                                     int B, int K, const int* exclude, long exclude_off, long idx_offset, float* out_scores,
                                     int* out_idx, unsigned long long* thr_key, int* flags, int* state, int* done, hipStream_t s, int inline_exact = 0,
                                     const float* tmax = nullptr, long ldt = 0);
hipError_t launch_topk_rescore(const TopkFilter& f, const float* q, const float* gallery, int D, const float* margin, int B, int K, long idx_offset,
                               float* out_scores, int* out_idx, int* flags, int* state, hipStream_t s);
// Fused sweep, final step: exact top-K of each query's candidate list -> out (idx = row + idx_offset; unfilled: -inf / -1).
// A query with an overflowed list (a count > cap) is not written in the first pass: its bound is raised to the K-th best of the
// stored candidates, its counts reset, flags[0] set, and the sweep + this kernel run again with pass = 1 (gated on
// flags[0]); queries that did not overflow get bound ~0 (reject all) for that pass.  An overflow in pass 1 sets *error_flag
// (host-mapped) and writes NaN scores / idx -1.
// Exact pass for the queries whose candidate lists overflowed (state[b] != 0; gated on flags[0]): every workgroup streams its share
// of the gallery through the SAME MFMA sequence as the sweep (bit-identical scores) into sorted wave lists, the last workgroup of
// a query merges the partial lists.  No capacity anywhere: the ranking stage is exact by construction.
hipError_t launch_rank_exact(const float* q, const void* gallery, int gallery_bf16, int B, long N, int D, int K, const int* state,
                             const unsigned long long* thr_key, const int* exclude, long exclude_off, long idx_offset,
                             unsigned long long* partial, int groups, int* done, float* out_scores, int* out_idx, const int* gate,
                             hipStream_t s);
hipError_t launch_topk_candidates(const TopkFilter& f, int B, int K, long idx_offset, float* out_scores, int* out_idx, int* flags,
                                  int* state, hipStream_t s);
// Merge R lists [R,B,K] (score, idx) -> [B,K]
hipError_t launch_topk_merge(const float* scores, const int* idx, float* out_scores, int* out_idx, int R, int B, int K,
                             hipStream_t s);

}  // namespace fern
