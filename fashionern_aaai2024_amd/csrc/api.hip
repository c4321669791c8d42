// C-ABI layer of libfern.so: context, weight repacking, and the per-model launch sequences
// (ViT / text towers, DVR fusion, index fusion, ranking).  See include/fern.h for the contract and the
// reference interface each entry point replaces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fern.h"
#include "kernels.h"

using namespace fern;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorInvalidValue ? FERN_ERR_ARG : FERN_ERR_HIP,                       \
                        std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)
#define FERN_TRY(expr)              \
    do {                            \
        int r_ = (expr);            \
        if (r_ != FERN_OK) return r_; \
    } while (0)

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<float> f;
    std::vector<int64_t> shape;
    size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

struct LinearW { const float* w = nullptr; const float* b = nullptr; int out = 0, in = 0; const unsigned short* wb = nullptr; /* bf16 copy of w (encoder blocks) */
                 const unsigned char* w8 = nullptr; const float* sw = nullptr; /* fp8 copy + per-output-channel scales */
                 const unsigned char* wm = nullptr; const unsigned char* swm = nullptr; long swm_rows = 0; /* MX fp8 copy + E8M0 block scales */ };
struct LNW { const float* g = nullptr; const float* b = nullptr; };
struct CombinerW {
    LinearW text, image, hidden;
    const float* w2 = nullptr;
    const float* b2 = nullptr;
    // query-side combiners (DVR.*): their hidden layer runs at M = batch (64) where an unsplit k chain of 4096 leaves 192 CUs
    // idle, so K is cut into 512-wide slices -- for EVERY batch size (kernels.h: GemmParams.ksplit): the split belongs to the
    // module, not to M.  The gallery-side combiner (M = thousands of rows per tile) stays unsplit.
    int ksplit = 1;
};
struct SRW {
    LinearW local, global;
    const float *bn13_mean = nullptr, *bn13_inv = nullptr, *bn13_beta = nullptr;   // per patch
    const float *bnd_scale = nullptr, *bnd_shift = nullptr;                         // per column
    const float *wc = nullptr, *bc = nullptr;
};
struct BertLayerW { LinearW qkv, attn_out, inter, out; LNW ln1, ln2; };
struct Clip4CirW { bool ready = false; int width = 0; LinearW text, image, comb, outl, hidden; const float* w2 = nullptr; const float* b2 = nullptr; };
struct FusionW {
    int parts = 0;      // FERN_PART_* bits that are finalised
    int D = 0;
    const float *cls = nullptr, *pos = nullptr, *type = nullptr;
    LNW emb_ln;
    BertLayerW layer[2];
    LinearW mha_q, mha_kv, mha_out;
    SRW sr[2];          // fern_sr_id
    CombinerW comb[4];  // fern_combiner_id
};
struct ClipBlockW { LNW ln1, ln2; LinearW qkv, out, fc, proj; };
struct ConvW { const float* w = nullptr; const float* b = nullptr; int cout = 0, k = 0; };   // BatchNorm folded; w [cout][k]
struct BottleneckW { ConvW c1, c2, c3, down; bool has_down = false; int stride = 1, cin = 0, planes = 0; };
struct ResNetW {
    ConvW stem1, stem2, stem3;
    int stem_c = 0;              // stem channel count padded to a multiple of 16 (RN50x4: 40 -> 48)
    std::vector<BottleneckW> blocks;
    const float* pos = nullptr;
    LinearW q, kv, cproj;
    const float* zeros = nullptr;
};
struct ClipW {
    bool ready = false;
    ResNetW res;
    fern_clip_config cfg{};
    const float *conv_w = nullptr, *cls = nullptr, *vpos = nullptr, *vproj_t = nullptr;
    LinearW conv_mx;                 // conv1 as a [width, 3 * patch * patch] linear layer: its block-scaled copy (FERN_PREC_MX8 patch embedding) and its bf16 copy (bf16-operand modes)
    LNW ln_pre, ln_post, ln_final;
    std::vector<ClipBlockW> vblocks, tblocks;
    const float *tok_emb = nullptr, *tpos = nullptr, *tproj_t = nullptr;
};

enum ProfKind { PROF_GEMM = 0, PROF_ATTN = 1, PROF_TOPK = 2, PROF_SWEEP = 3, PROF_STAGE = 4 };
struct ProfRec { hipEvent_t a, b; int kind; double work; int m, n, k, tag; int dispatches;
                 std::vector<hipEvent_t> kev; /* kernel-precise (start, stop) pairs of the launch's dispatches (kernels.h: LaunchTimer); empty: a, b are stream events */
                 /* span records (the ranking stage, round 5): kev holds EVERY dispatch of the stage in launch order; the stage's time is first
                    dispatch begin -> last dispatch end (launch boundaries between them included), the dispatches [sweep_lo, sweep_hi) are its
                    full-gallery sweep(s), `work` their bytes */
                 bool span = false; int sweep_lo = 0, sweep_hi = 0;
                 double extra_flops = 0.0; /* run_gemm_b_pair: the flops of `work` that belong to the second (bf16) problem */
                 double extra_alg_bytes = 0.0; /* a GEMM PAIR record (run_gemm_pair): the second problem's algorithmic bytes (m, n, k are the first's) */ };

namespace fern {
thread_local LaunchTimer* g_launch_timer = nullptr;
hipEvent_t launch_timer_event() {
    LaunchTimer* t = g_launch_timer;
    hipEvent_t e = nullptr;
    if (t && t->pool && !t->pool->empty()) { e = t->pool->back(); t->pool->pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
}
}  // namespace fern

struct fern_ctx {
    int device = 0;
    std::map<std::string, HostTensor> host;
    // device weight buffers, per weight group: re-finalising a group frees its previous generation
    enum { G_CLIP = 0, G_DVR, G_TARGET_SR, G_TARGET_COMBINER, G_CLIP4CIR, G_COUNT };
    std::vector<void*> owned[G_COUNT];
    int cur_group = G_CLIP;          // group the upload helpers currently allocate into
    // forks point into the parent's buffers: `generation` counts the parent's re-finalisations, a fork remembers the value
    // it was created at and every weight-reading entry point refuses to run on a stale fork
    unsigned generation = 0;
    const fern_ctx* parent = nullptr;
    unsigned parent_generation = 0;
    int* tok_flag = nullptr;         // host-mapped: set by the text embedding kernel when a token id is out of range
    FusionW fusion;
    ClipW clip;
    Clip4CirW c4c;
    int precision = FERN_PREC_FP32;  // operand precision of the CLIP towers' token-level GEMMs (fern_set_precision)
    int rank_strategy = 0;           // fern_rank_set_strategy (FERN_RANK_AUTO)
    bool f32x3 = false;              // FERN_PREC_F32X3: `precision` stays FP32 (same buffers, same code paths), GEMMs run split (run_gemm)
    // workspace arena (bump allocator; blocks are consolidated at the start of the next op)
    struct Block { char* p; size_t cap; };
    std::vector<Block> blocks;
    size_t used = 0;                 // offset into blocks.back()
    size_t op_total = 0;
    // bumped whenever workspace memory handed out earlier is FREED (the consolidation in ws_begin): addresses baked into a
    // captured hipGraph of this context are dead from then on -- fern_ws_generation lets the owner of the graph notice
    unsigned long long ws_generation = 0;
    // profiling
    bool prof_on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
    LaunchTimer timer;
};

static int ws_begin(fern_ctx* c, hipStream_t s) {
    if (c->blocks.size() > 1) {      // grew during the previous op: merge into one block (outside capture only)
        size_t total = 0;
        for (auto& b : c->blocks) total += b.cap;
        HIP_TRY(hipStreamSynchronize(s));
        for (auto& b : c->blocks) HIP_TRY(hipFree(b.p));
        c->blocks.clear();
        ++c->ws_generation;
        char* p = nullptr;
        HIP_TRY(hipMalloc(&p, total));
        c->blocks.push_back({p, total});
    }
    c->used = 0;
    return FERN_OK;
}
static int ws_alloc(fern_ctx* c, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (c->blocks.empty() || c->used + bytes > c->blocks.back().cap) {
        size_t cap = bytes;
        if (!c->blocks.empty()) cap = std::max(bytes, c->blocks.back().cap);   // geometric growth
        cap = std::max(cap, (size_t)64 << 20);
        char* p = nullptr;
        HIP_TRY(hipMalloc(&p, cap));
        c->blocks.push_back({p, cap});
        c->used = 0;
    }
    *out = c->blocks.back().p + c->used;
    c->used += bytes;
    return FERN_OK;
}
template <class T>
static int ws_get(fern_ctx* c, size_t count, T** out) {
    void* p = nullptr;
    FERN_TRY(ws_alloc(c, count * sizeof(T), &p));
    *out = reinterpret_cast<T*>(p);
    return FERN_OK;
}

// ---- profiling hooks ----------------------------------------------------------------------------
// GEMM and attention launches (kinds whose roofline is a per-KERNEL figure): timed by the dispatches' own timestamps
// (the sweep kernel of the ranking stage too; the STAGE around it -- sample pass, bound, sweep, select, exact gate, launch boundaries
// included -- is ONE stream-marker interval, PROF_STAGE)
static bool prof_kernel_precise(int kind) { return kind == PROF_GEMM || kind == PROF_ATTN || kind == PROF_SWEEP; }
static int prof_open(fern_ctx* c, int kind, double work, hipStream_t s, int* slot, int m = 0, int n = 0, int k = 0, int tag = 0) {
    *slot = -1;
    if (!c->prof_on) return FERN_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (prof_kernel_precise(kind) && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
        c->recs.push_back({nullptr, nullptr, kind, work, m, n, k, tag, 1, {}});
        *slot = (int)c->recs.size() - 1;
        c->timer.events.clear();
        c->timer.pool = &c->ev_pool;
        g_launch_timer = &c->timer;      // armed: the launch's dispatches go through hipExtLaunchKernelGGL (kernels.h: FERN_LAUNCH)
        return FERN_OK;
    }
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) {
        if (!c->ev_pool.empty()) { ev[i] = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else HIP_TRY(hipEventCreate(&ev[i]));
    }
    HIP_TRY(hipEventRecord(ev[0], s));
    c->recs.push_back({ev[0], ev[1], kind, work, m, n, k, tag, 1, {}});
    *slot = (int)c->recs.size() - 1;
    return FERN_OK;
}
static int prof_close(fern_ctx* c, int slot, hipStream_t s) {
    if (slot < 0) return FERN_OK;
    if (!c->recs[slot].a) {              // kernel-precise record: collect the event pairs of its dispatches, disarm
        g_launch_timer = nullptr;
        c->recs[slot].kev.swap(c->timer.events);
        return FERN_OK;
    }
    HIP_TRY(hipEventRecord(c->recs[slot].b, s));
    return FERN_OK;
}

// A launch that failed after prof_open: disarm the timer, hand the events of the dispatches that did go out back to the pool and
// drop the record -- left in place it would count as a 0 ms launch carrying its full flops.  Records are appended and
// closed in order on one thread, so an open record is always the last one.
static void prof_abort(fern_ctx* c, int slot) {
    g_launch_timer = nullptr;
    for (hipEvent_t e : c->timer.events) c->ev_pool.push_back(e);
    c->timer.events.clear();
    if (slot < 0 || slot != (int)c->recs.size() - 1) return;
    ProfRec& r = c->recs[slot];
    if (r.a) c->ev_pool.push_back(r.a);
    if (r.b) c->ev_pool.push_back(r.b);
    c->recs.pop_back();
}
#define HIP_TRY_PROF(le, c, slot)                  \
    do {                                            \
        const hipError_t le__ = (le);               \
        if (le__ != hipSuccess) prof_abort(c, slot); \
        HIP_TRY(le__);                              \
    } while (0)

// The ranking stage timed by its dispatches' own timestamps (round 5): while a StageTimer is live every launch of the stage goes through
// FERN_LAUNCH with an event pair; commit() turns them into ONE span record.  A stage that fails (or a stream being captured) leaves
// nothing behind.  [The stream-marker interval of rounds 3-4 charged two marker packets + dispatch latencies, ~8 us, to a stage that is
// now ~46 us of kernels.]
struct StageTimer {
    fern_ctx* c;
    bool on = false, done = false;
    int sweep_lo = 0, sweep_hi = 0;
    double sweep_bytes = 0;
    StageTimer(fern_ctx* ctx, hipStream_t s) : c(ctx) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (c->prof_on && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            c->timer.events.clear();
            c->timer.pool = &c->ev_pool;
            g_launch_timer = &c->timer;
            on = true;
        }
    }
    void sweep_begin() { if (on) sweep_lo = (int)c->timer.events.size() / 2; }
    void sweep_end(double bytes) { if (on) { sweep_hi = (int)c->timer.events.size() / 2; sweep_bytes += bytes; } }
    void commit(int m, int n, int k) {
        if (!on) return;
        g_launch_timer = nullptr;
        done = true;
        if (c->timer.events.size() < 2) return;
        ProfRec r{nullptr, nullptr, PROF_STAGE, sweep_bytes, m, n, k, 0, 1, {}};
        r.span = true; r.sweep_lo = sweep_lo; r.sweep_hi = sweep_hi;
        r.kev.swap(c->timer.events);
        c->recs.push_back(std::move(r));
    }
    ~StageTimer() {
        if (on && !done) {
            g_launch_timer = nullptr;
            for (hipEvent_t e : c->timer.events) c->ev_pool.push_back(e);
            c->timer.events.clear();
        }
    }
};

static int run_gemm(fern_ctx* c, const GemmParams& p, hipStream_t s, int kind = PROF_GEMM, double work = -1.0) {
    int slot;
    FERN_TRY(prof_open(c, kind, work >= 0 ? work : 2.0 * p.M * (double)p.N * p.K, s, &slot, p.M, p.N, p.K, p.epi));
    hipError_t le;
    if (c->f32x3 && kind == PROF_GEMM && p.w_sample <= 1) {      // FERN_PREC_F32X3: never the ranking stage (PROF_SWEEP / sample pass)
        GemmParams q = p;
        q.split = 3;                                             // launch_gemm falls back to the fp32 kernels where the split family does not apply
        le = launch_gemm(q, s);
    } else {
        le = launch_gemm(p, s);
    }
    HIP_TRY_PROF(le, c, slot);                                   // (a failed launch must not leave the timer armed, nor a 0 ms record behind)
    if (slot >= 0) c->recs[slot].dispatches = gemm_last_dispatches();
    return prof_close(c, slot, s);
}
// Two plain GEMMs of the fp32 data flow in ONE launch where the family can (gemm.hip: launch_gemm_pair; two launches otherwise): the
// image tower's GEMM of a layer and the text tower's GEMM of the same kind.  One profile record for both (shape = the first's, the
// flops of both): the launch is what the roofline prices.
static int run_gemm_pair(fern_ctx* c, const GemmParams& p1, const GemmParams& p2, hipStream_t s) {
    int slot;
    FERN_TRY(prof_open(c, PROF_GEMM, 2.0 * p1.M * (double)p1.N * p1.K + 2.0 * p2.M * (double)p2.N * p2.K, s, &slot, p1.M, p1.N, p1.K, 50 + p1.epi));
    GemmParams a = p1, b = p2;
    if (c->f32x3) { a.split = 3; b.split = 3; }
    const hipError_t le = launch_gemm_pair(a, b, s);
    HIP_TRY_PROF(le, c, slot);
    if (slot >= 0) {
        c->recs[slot].dispatches = gemm_last_dispatches();
        c->recs[slot].extra_alg_bytes = 4.0 * ((double)p2.M * p2.K + (double)p2.N * p2.K + (double)p2.M * p2.N);
    }
    return prof_close(c, slot, s);
}
static GemmParams gemm_desc(const float* A, long lda, const LinearW& L, float* C, long ldc, int M, int epi) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = L.w; p.ldw = L.in; p.bias = L.b; p.C = C; p.ldc = ldc;
    p.M = M; p.N = L.out; p.K = L.in; p.epi = epi; p.aload = ALOAD_PLAIN;
    return p;
}
// bf16-operand GEMM of the encoder "perf mode": A is a bf16 activation buffer, the weight is the layer's bf16 copy
static GemmParams gemm_desc_b(const unsigned short* A, long lda, const LinearW& L, void* C, long ldc, int M, int epi, bool out_bf16) {
    GemmParams p{};
    p.Ab = A; p.lda = lda; p.Wb = L.wb; p.ldw = L.in; p.bias = L.b; p.C = reinterpret_cast<float*>(C); p.ldc = ldc;
    p.M = M; p.N = L.out; p.K = L.in; p.epi = epi; p.aload = ALOAD_PLAIN; p.out_bf16 = out_bf16 ? 1 : 0;
    return p;
}
// fp8-operand GEMM: A8 [M, K] e4m3fn with per-row scales sa, the layer's fp8 weight copy with per-channel scales
static GemmParams gemm_desc_f8(const unsigned char* A8, const float* sa, long lda, const LinearW& L, void* C, long ldc, int M, int epi,
                               bool out_bf16) {
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A8); p.lda = lda; p.Wb = reinterpret_cast<const unsigned short*>(L.w8); p.ldw = L.in;
    p.bias = L.b; p.C = reinterpret_cast<float*>(C); p.ldc = ldc; p.M = M; p.N = L.out; p.K = L.in; p.epi = epi; p.aload = ALOAD_PLAIN;
    p.out_bf16 = out_bf16 ? 1 : 0; p.fp8 = 1; p.scale_a = sa; p.scale_w = L.sw;
    return p;
}
// block-scaled (MX) fp8 GEMM: A8 [M, K] e4m3fn with E8M0 block scales sa (srows rows), the layer's MX weight copy
static GemmParams gemm_desc_mx(const unsigned char* A8, const unsigned char* sa, long srows, long lda, const LinearW& L, void* C, long ldc, int M,
                               int epi, bool out_bf16) {
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A8); p.lda = lda; p.Wb = reinterpret_cast<const unsigned short*>(L.wm); p.ldw = L.in;
    p.bias = L.b; p.C = reinterpret_cast<float*>(C); p.ldc = ldc; p.M = M; p.N = L.out; p.K = L.in; p.epi = epi; p.aload = ALOAD_PLAIN;
    p.out_bf16 = out_bf16 ? 1 : 0; p.fp8 = 2; p.mxa = sa; p.mxa_rows = srows; p.mxw = L.swm; p.mxw_rows = L.swm_rows;
    return p;
}
static int run_gemm_b(fern_ctx* c, const GemmParams& p, hipStream_t s) {
    int slot;
    FERN_TRY(prof_open(c, PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, s, &slot, p.M, p.N, p.K, (p.fp8 == 2 ? 300 : p.fp8 ? 200 : 100) + p.epi));
    const hipError_t le = launch_gemm_bf16(p, s);
    HIP_TRY_PROF(le, c, slot);
    return prof_close(c, slot, s);
}
// The image tower's block-scaled GEMM of a layer and the text tower's bf16 GEMM of the same kind in ONE launch where that wins
// (gemm_bf16.hip: launch_gemm_mxbf_pair; two launches otherwise).  One profile record, as run_gemm_pair.
static int run_gemm_b_pair(fern_ctx* c, const GemmParams& p1, const GemmParams& p2, hipStream_t s) {
    int slot;
    FERN_TRY(prof_open(c, PROF_GEMM, 2.0 * p1.M * (double)p1.N * p1.K + 2.0 * p2.M * (double)p2.N * p2.K, s, &slot, p1.M, p1.N, p1.K, 350 + p1.epi));
    const hipError_t le = launch_gemm_mxbf_pair(p1, p2, s);
    HIP_TRY_PROF(le, c, slot);
    if (slot >= 0) {
        c->recs[slot].dispatches = gemm_bf16_last_dispatches();
        c->recs[slot].extra_flops = 2.0 * p2.M * (double)p2.N * p2.K;
    }
    return prof_close(c, slot, s);
}
static int run_attention(fern_ctx* c, const AttnParams& a, hipStream_t s) {
    int slot;
    FERN_TRY(prof_open(c, PROF_ATTN, 4.0 * a.batch * a.heads * (double)a.s_q * a.s_k * a.hd, s, &slot, a.batch * a.heads, a.s_q, a.hd, a.causal));
    const hipError_t le = launch_attention(a, s);
    HIP_TRY_PROF(le, c, slot);
    return prof_close(c, slot, s);
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static int make_bf16(fern_ctx* c, LinearW* L);

static int upload(fern_ctx* c, const float* h, size_t n, const float** out) {
    float* d = nullptr;
    HIP_TRY(hipMalloc(&d, std::max<size_t>(n, 4) * sizeof(float)));
    HIP_TRY(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    c->owned[c->cur_group].push_back(d);
    *out = d;
    return FERN_OK;
}
static int need(fern_ctx* c, const std::string& key, std::vector<int64_t> shape, const HostTensor** out) {
    auto it = c->host.find(key);
    if (it == c->host.end()) return fail(FERN_ERR_STATE, "missing weight: " + key);
    if (!shape.empty() && it->second.shape != shape) {
        std::string got, want;
        for (auto s : it->second.shape) got += std::to_string(s) + ",";
        for (auto s : shape) want += std::to_string(s) + ",";
        return fail(FERN_ERR_STATE, "weight " + key + " has shape [" + got + "] expected [" + want + "]");
    }
    *out = &it->second;
    return FERN_OK;
}
static int up_key(fern_ctx* c, const std::string& key, std::vector<int64_t> shape, const float** out) {
    const HostTensor* t;
    FERN_TRY(need(c, key, shape, &t));
    return upload(c, t->f.data(), t->f.size(), out);
}
static int up_linear(fern_ctx* c, const std::string& prefix, int out_f, int in_f, LinearW* L) {
    FERN_TRY(up_key(c, prefix + ".weight", {out_f, in_f}, &L->w));
    FERN_TRY(up_key(c, prefix + ".bias", {out_f}, &L->b));
    L->out = out_f; L->in = in_f;
    return FERN_OK;
}
static int up_ln(fern_ctx* c, const std::string& prefix, int n, LNW* L) {
    FERN_TRY(up_key(c, prefix + ".weight", {n}, &L->g));
    return up_key(c, prefix + ".bias", {n}, &L->b);
}
// concat several [rows_i, in] weights (+ biases) into one packed Linear
static int up_packed(fern_ctx* c, const std::vector<std::string>& prefixes, int out_each, int in_f, LinearW* L) {
    std::vector<float> w, b;
    for (auto& p : prefixes) {
        const HostTensor *tw, *tb;
        FERN_TRY(need(c, p + ".weight", {out_each, in_f}, &tw));
        FERN_TRY(need(c, p + ".bias", {out_each}, &tb));
        w.insert(w.end(), tw->f.begin(), tw->f.end());
        b.insert(b.end(), tb->f.begin(), tb->f.end());
    }
    FERN_TRY(upload(c, w.data(), w.size(), &L->w));
    FERN_TRY(upload(c, b.data(), b.size(), &L->b));
    L->out = out_each * (int)prefixes.size(); L->in = in_f;
    return FERN_OK;
}
static int up_transposed(fern_ctx* c, const std::string& key, int rows, int cols, const float** out) {   // [rows, cols] -> [cols, rows]
    const HostTensor* t;
    FERN_TRY(need(c, key, {rows, cols}, &t));
    std::vector<float> tr((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int q = 0; q < cols; ++q) tr[(size_t)q * rows + r] = t->f[(size_t)r * cols + q];
    return upload(c, tr.data(), tr.size(), out);
}

static int up_sr(fern_ctx* c, const std::string& p, int D, SRW* W) {
    FERN_TRY(up_linear(c, p + ".embedding_local.0", D, D, &W->local));
    FERN_TRY(up_linear(c, p + ".embedding_global.0", D, D, &W->global));
    const HostTensor *w, *b, *rm, *rv;
    // BatchNorm1d(13) over the PATCH axis (fusion_model.py:117-123 applied to [n,13,D])
    FERN_TRY(need(c, p + ".embedding_local.1.weight", {13}, &w));
    FERN_TRY(need(c, p + ".embedding_local.1.bias", {13}, &b));
    FERN_TRY(need(c, p + ".embedding_local.1.running_mean", {13}, &rm));
    FERN_TRY(need(c, p + ".embedding_local.1.running_var", {13}, &rv));
    std::vector<float> inv(13);
    for (int i = 0; i < 13; ++i) inv[i] = w->f[i] / std::sqrt(rv->f[i] + 1e-5f);
    FERN_TRY(upload(c, rm->f.data(), 13, &W->bn13_mean));
    FERN_TRY(upload(c, inv.data(), 13, &W->bn13_inv));
    FERN_TRY(upload(c, b->f.data(), 13, &W->bn13_beta));
    // BatchNorm1d(D) over the feature axis, folded to scale/shift
    FERN_TRY(need(c, p + ".embedding_global.1.weight", {D}, &w));
    FERN_TRY(need(c, p + ".embedding_global.1.bias", {D}, &b));
    FERN_TRY(need(c, p + ".embedding_global.1.running_mean", {D}, &rm));
    FERN_TRY(need(c, p + ".embedding_global.1.running_var", {D}, &rv));
    std::vector<float> sc(D), sh(D);
    for (int i = 0; i < D; ++i) {
        sc[i] = w->f[i] / std::sqrt(rv->f[i] + 1e-5f);
        sh[i] = b->f[i] - rm->f[i] * sc[i];
    }
    FERN_TRY(upload(c, sc.data(), D, &W->bnd_scale));
    FERN_TRY(upload(c, sh.data(), D, &W->bnd_shift));
    FERN_TRY(up_key(c, p + ".embedding_common.weight", {1, D}, &W->wc));
    return up_key(c, p + ".embedding_common.bias", {1}, &W->bc);
}
static int up_combiner(fern_ctx* c, const std::string& p, int D, CombinerW* W) {
    // CombinerSimple(clip_feature_dim=D, projection_dim=Pj, hidden_dim=Hd) (fusion_model.py:63-84); ERN uses Pj=4D, Hd=8D
    const HostTensor *tp, *hw;
    FERN_TRY(need(c, p + ".text_projection_layer.0.weight", {}, &tp));
    FERN_TRY(need(c, p + ".dynamic_scalar.0.weight", {}, &hw));
    if (tp->shape.size() != 2 || hw->shape.size() != 2) return fail(FERN_ERR_STATE, "combiner weights must be 2-D: " + p);
    const int Pj = (int)tp->shape[0], Hd = (int)hw->shape[0];
    if (Pj % 4 || (2 * Pj) % 32 || Hd <= 0) return fail(FERN_ERR_ARG, "combiner: projection_dim must be a multiple of 16: " + p);
    FERN_TRY(up_linear(c, p + ".text_projection_layer.0", Pj, D, &W->text));
    FERN_TRY(up_linear(c, p + ".image_projection_layer.0", Pj, D, &W->image));
    FERN_TRY(up_linear(c, p + ".dynamic_scalar.0", Hd, 2 * Pj, &W->hidden));
    FERN_TRY(up_key(c, p + ".dynamic_scalar.3.weight", {1, Hd}, &W->w2));
    return up_key(c, p + ".dynamic_scalar.3.bias", {1}, &W->b2);
}

// Start (re-)finalising one weight group: everything that may still read the previous generation is drained, the old
// buffers are released, and forks created before this point become stale (they hold pointers into the freed memory).
static int begin_group(fern_ctx* c, int group) {
    if (c->parent) return fail(FERN_ERR_STATE, "weights are finalised on the root context, not on a fork");
    c->cur_group = group;
    if (!c->owned[group].empty()) {
        HIP_TRY(hipDeviceSynchronize());
        for (void* p : c->owned[group]) HIP_TRY(hipFree(p));
        c->owned[group].clear();
    }
    c->generation++;
    return FERN_OK;
}
static int check_token_flag(fern_ctx* c, const char* fn);
// Weight-reading entry points call this first: a fork made before the parent's last re-finalisation must not run.
static int check_fresh(fern_ctx* c, const char* fn) {
    if (c->parent && c->parent->generation != c->parent_generation)
        return fail(FERN_ERR_STATE, std::string(fn) + ": this fork predates the parent's last fern_finalize_*: its weight pointers are stale "
                                                       "(fork again after loading weights)");
    return FERN_OK;
}

// ------------------------------------------------------------------------------------------------
// exported: lifetime, weights
// ------------------------------------------------------------------------------------------------
extern "C" int fern_abi_version(void) { return FERN_ABI_VERSION; }
extern "C" const char* fern_last_error(void) { return g_last_error.c_str(); }

extern "C" int fern_ctx_create(int device, fern_ctx** out) {
    if (!out) return fail(FERN_ERR_ARG, "fern_ctx_create: out is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(FERN_ERR_ARG, "fern_ctx_create: no such device " + std::to_string(device));
    HIP_TRY(hipSetDevice(device));
    auto* c = new fern_ctx();
    c->device = device;
    *out = c;
    return FERN_OK;
}
// A fork shares the parent's finalised device weights (read-only) but has its own workspace, tuner-independent launch
// state and profiling records, so several forks can run concurrently on different HIP streams of the same device.
extern "C" int fern_ctx_fork(fern_ctx* parent, fern_ctx** out) {
    if (!parent || !out) return fail(FERN_ERR_ARG, "fern_ctx_fork: NULL argument");
    auto* c = new fern_ctx();
    c->device = parent->device;
    if (parent->parent) return fail(FERN_ERR_ARG, "fern_ctx_fork: fork the root context, not a fork");
    c->parent = parent;
    c->parent_generation = parent->generation;
    c->fusion = parent->fusion;      // pointers into the parent's `owned` buffers; the parent must outlive its forks
    c->clip = parent->clip;
    c->c4c = parent->c4c;
    c->precision = parent->precision;
    c->rank_strategy = parent->rank_strategy;
    c->f32x3 = parent->f32x3;
    *out = c;
    return FERN_OK;
}

extern "C" int fern_ctx_destroy(fern_ctx* c) {
    if (!c) return FERN_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& grp : c->owned)
        for (void* p : grp) (void)hipFree(p);
    if (c->tok_flag) (void)hipHostFree(c->tok_flag);
    for (auto& b : c->blocks) (void)hipFree(b.p);
    for (auto& r : c->recs) {
        if (r.a) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (hipEvent_t e : r.kev) (void)hipEventDestroy(e);
    }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
    return FERN_OK;
}
extern "C" int fern_sync(fern_ctx* c, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_sync: ctx is NULL");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return check_token_flag(c, "fern_sync");
}

extern "C" int fern_load_tensor(fern_ctx* c, const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape) {
    if (!c || !key) return fail(FERN_ERR_ARG, "fern_load_tensor: NULL argument");
    if (ndim < 0 || ndim > 8 || (ndim > 0 && !shape)) return fail(FERN_ERR_ARG, "fern_load_tensor: bad ndim/shape");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(FERN_ERR_ARG, "fern_load_tensor: negative dimension");
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    if (n && !host_ptr) return fail(FERN_ERR_ARG, "fern_load_tensor: host_ptr is NULL");
    t.f.resize(n);
    if (dtype == FERN_F32) std::memcpy(t.f.data(), host_ptr, n * sizeof(float));
    else if (dtype == FERN_I64) for (size_t i = 0; i < n; ++i) t.f[i] = (float)((const int64_t*)host_ptr)[i];
    else return fail(FERN_ERR_ARG, "fern_load_tensor: unsupported dtype");
    c->host[key] = std::move(t);
    return FERN_OK;
}

extern "C" int fern_finalize_fusion(fern_ctx* c, int D, int parts) {
    if (!c) return fail(FERN_ERR_ARG, "fern_finalize_fusion: ctx is NULL");
    if (D <= 0 || D % 32 || D > 768) return fail(FERN_ERR_ARG, "fern_finalize_fusion: feature_dim must be a multiple of 32, <= 768");
    if (parts <= 0 || (parts & ~FERN_PART_ALL)) return fail(FERN_ERR_ARG, "fern_finalize_fusion: bad parts mask");
    HIP_TRY(hipSetDevice(c->device));
    FusionW& F = c->fusion;
    if (F.parts && F.D != D) return fail(FERN_ERR_ARG, "fern_finalize_fusion: feature_dim differs from the already finalised parts");
    F.D = D;
    if (parts & FERN_PART_DVR) {
        F.parts &= ~FERN_PART_DVR;
        FERN_TRY(begin_group(c, fern_ctx::G_DVR));
        const std::string tl = "DVR.transformer_layer";
        const std::string bm = tl + ".bert_encoder.bert_model";
        if (c->host.count(tl + ".cls_token")) FERN_TRY(up_key(c, tl + ".cls_token", {1, 1, D}, &F.cls));
        else {   // absent in GPU-trained checkpoints: `nn.Parameter(...).to(device)` is not registered (fusion_model.py:185)
            std::vector<float> z(D, 0.f);
            FERN_TRY(upload(c, z.data(), D, &F.cls));
        }
        const HostTensor* pos;
        FERN_TRY(need(c, bm + ".embeddings.position_embeddings.weight", {}, &pos));
        if (pos->shape.size() != 2 || pos->shape[1] != D || pos->shape[0] < 96) return fail(FERN_ERR_STATE, "position_embeddings has the wrong shape");
        FERN_TRY(upload(c, pos->f.data(), pos->f.size(), &F.pos));
        FERN_TRY(up_key(c, bm + ".embeddings.token_type_embeddings.weight", {2, D}, &F.type));
        FERN_TRY(up_ln(c, bm + ".embeddings.LayerNorm", D, &F.emb_ln));
        for (int i = 0; i < 2; ++i) {
            const std::string lp = bm + ".encoder.layer." + std::to_string(i);
            BertLayerW& L = F.layer[i];
            FERN_TRY(up_packed(c, {lp + ".attention.self.query", lp + ".attention.self.key", lp + ".attention.self.value"}, D, D, &L.qkv));
            FERN_TRY(up_linear(c, lp + ".attention.output.dense", D, D, &L.attn_out));
            FERN_TRY(up_ln(c, lp + ".attention.output.LayerNorm", D, &L.ln1));
            const HostTensor* iw;
            FERN_TRY(need(c, lp + ".intermediate.dense.weight", {}, &iw));
            const int inter = (int)iw->shape[0];
            if (inter % 32) return fail(FERN_ERR_ARG, "BERT intermediate size must be a multiple of 32");
            FERN_TRY(up_linear(c, lp + ".intermediate.dense", inter, D, &L.inter));
            FERN_TRY(up_linear(c, lp + ".output.dense", D, inter, &L.out));
            FERN_TRY(up_ln(c, lp + ".output.LayerNorm", D, &L.ln2));
            // bf16 copies for the reduced-precision modes (the fusion BERT blocks follow the towers' bf16 recipe)
            FERN_TRY(make_bf16(c, &L.qkv));
            FERN_TRY(make_bf16(c, &L.attn_out));
            FERN_TRY(make_bf16(c, &L.inter));
            FERN_TRY(make_bf16(c, &L.out));
        }
        HIP_TRY(hipStreamSynchronize(nullptr));
        {   // nn.MultiheadAttention packed in-proj: rows [0,D) = q, [D,3D) = k,v
            const float *w, *b;
            FERN_TRY(up_key(c, "DVR.MR_component.in_proj_weight", {3 * D, D}, &w));
            FERN_TRY(up_key(c, "DVR.MR_component.in_proj_bias", {3 * D}, &b));
            F.mha_q = {w, b, D, D};
            F.mha_kv = {w + (size_t)D * D, b + D, 2 * D, D};
            FERN_TRY(up_linear(c, "DVR.MR_component.out_proj", D, D, &F.mha_out));
        }
        FERN_TRY(up_sr(c, "DVR.SR_module", D, &F.sr[FERN_SR_DVR]));
        FERN_TRY(up_combiner(c, "DVR.combiner_global", D, &F.comb[FERN_COMBINER_DVR_GLOBAL]));
        FERN_TRY(up_combiner(c, "DVR.combiner_local", D, &F.comb[FERN_COMBINER_DVR_LOCAL]));
        FERN_TRY(up_combiner(c, "DVR.combiner", D, &F.comb[FERN_COMBINER_DVR_FINAL]));
        for (int w : {FERN_COMBINER_DVR_GLOBAL, FERN_COMBINER_DVR_LOCAL, FERN_COMBINER_DVR_FINAL}) {
            CombinerW& Cw = F.comb[w];      // a function of the layer's (N, K) only
            Cw.ksplit = (Cw.hidden.in >= 2048 && Cw.hidden.in % 512 == 0 && Cw.hidden.out % 32 == 0) ? Cw.hidden.in / 512 : 1;
        }
        F.parts |= FERN_PART_DVR;
    }
    if (parts & FERN_PART_TARGET_SR) {
        F.parts &= ~FERN_PART_TARGET_SR;
        FERN_TRY(begin_group(c, fern_ctx::G_TARGET_SR));
        FERN_TRY(up_sr(c, "SR_module", D, &F.sr[FERN_SR_TARGET]));
        F.parts |= FERN_PART_TARGET_SR;
    }
    if (parts & FERN_PART_TARGET_COMBINER) {
        F.parts &= ~FERN_PART_TARGET_COMBINER;
        FERN_TRY(begin_group(c, fern_ctx::G_TARGET_COMBINER));
        FERN_TRY(up_combiner(c, "Combiner_module", D, &F.comb[FERN_COMBINER_TARGET]));
        F.parts |= FERN_PART_TARGET_COMBINER;
    }
    return FERN_OK;
}

// bf16 (round-to-nearest-even) device copy of an uploaded fp32 weight matrix, for the perf-mode GEMMs
static int make_bf16(fern_ctx* c, LinearW* L) {
    const size_t n = (size_t)L->out * L->in;
    unsigned short* d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(unsigned short)));
    c->owned[c->cur_group].push_back(d);
    HIP_TRY(launch_f32_to_bf16(L->w, d, (long)n, nullptr));
    L->wb = d;
    return FERN_OK;
}

// fp8 (e4m3fn) device copy with one scale per output channel (row of the [out, in] matrix)
static int make_fp8(fern_ctx* c, LinearW* L) {
    if (L->in % 64 || L->in > 4096) return FERN_OK;      // shape outside the fp8 GEMM: the layer keeps bf16 / fp32 only
    unsigned char* d = nullptr;
    float* sw = nullptr;
    HIP_TRY(hipMalloc(&d, (size_t)L->out * L->in));
    c->owned[c->cur_group].push_back(d);
    HIP_TRY(hipMalloc(&sw, (size_t)L->out * sizeof(float)));
    c->owned[c->cur_group].push_back(sw);
    HIP_TRY(launch_quantize_rows_fp8(nullptr, L->w, L->in, d, L->in, sw, L->out, L->in, nullptr));
    L->w8 = d;
    L->sw = sw;
    return FERN_OK;
}

// MX (block-scaled) e4m3fn device copy: one E8M0 byte per (output channel, 32 consecutive k), kernels.h: mx_scale_offset layout
static int make_mx8(fern_ctx* c, LinearW* L) {
    if (L->in % 128 || L->in > 4096) return FERN_OK;     // shape outside the MX GEMM
    unsigned char *d = nullptr, *sc = nullptr;
    HIP_TRY(hipMalloc(&d, (size_t)L->out * L->in));
    c->owned[c->cur_group].push_back(d);
    HIP_TRY(hipMalloc(&sc, (size_t)L->out * (L->in / 32)));
    c->owned[c->cur_group].push_back(sc);
    HIP_TRY(launch_quantize_mx8(nullptr, L->w, L->in, d, L->in, sc, L->out, L->out, L->in, nullptr));
    L->wm = d;
    L->swm = sc;
    L->swm_rows = L->out;
    return FERN_OK;
}

static int up_clip_block(fern_ctx* c, const std::string& p, int width, int mlp, ClipBlockW* B) {
    FERN_TRY(up_ln(c, p + ".ln_1", width, &B->ln1));
    FERN_TRY(up_key(c, p + ".attn.in_proj_weight", {3 * width, width}, &B->qkv.w));
    FERN_TRY(up_key(c, p + ".attn.in_proj_bias", {3 * width}, &B->qkv.b));
    B->qkv.out = 3 * width; B->qkv.in = width;
    FERN_TRY(up_linear(c, p + ".attn.out_proj", width, width, &B->out));
    FERN_TRY(up_ln(c, p + ".ln_2", width, &B->ln2));
    FERN_TRY(up_linear(c, p + ".mlp.c_fc", mlp, width, &B->fc));
    FERN_TRY(up_linear(c, p + ".mlp.c_proj", width, mlp, &B->proj));
    FERN_TRY(make_bf16(c, &B->qkv));
    FERN_TRY(make_bf16(c, &B->out));
    FERN_TRY(make_bf16(c, &B->fc));
    FERN_TRY(make_bf16(c, &B->proj));
    FERN_TRY(make_fp8(c, &B->qkv));
    FERN_TRY(make_fp8(c, &B->out));
    FERN_TRY(make_fp8(c, &B->fc));
    FERN_TRY(make_fp8(c, &B->proj));
    FERN_TRY(make_mx8(c, &B->qkv));
    FERN_TRY(make_mx8(c, &B->out));
    FERN_TRY(make_mx8(c, &B->fc));
    return make_mx8(c, &B->proj);
}

// conv (no bias) + BatchNorm(eval) -> [cout_pad][kh*kw*cin_pad] weights in (ky, kx, ci) order (or the original (ci, ky, kx)
// order for the direct stem kernel) with the BN scale folded in, and a bias vector; padded rows / channels are zero.
static int up_conv_bn(fern_ctx* c, const std::string& conv, const std::string& bn, int cout, int cin, int ks, int cout_pad, int cin_pad,
                      bool tap_major, ConvW* out) {
    const HostTensor *w, *g, *b, *rm, *rv;
    FERN_TRY(need(c, conv + ".weight", {cout, cin, ks, ks}, &w));
    FERN_TRY(need(c, bn + ".weight", {cout}, &g));
    FERN_TRY(need(c, bn + ".bias", {cout}, &b));
    FERN_TRY(need(c, bn + ".running_mean", {cout}, &rm));
    FERN_TRY(need(c, bn + ".running_var", {cout}, &rv));
    const int K = ks * ks * cin_pad;
    std::vector<float> wf((size_t)cout_pad * K, 0.f), bf(cout_pad, 0.f);
    for (int o = 0; o < cout; ++o) {
        const float sc = g->f[o] / std::sqrt(rv->f[o] + 1e-5f);
        bf[o] = b->f[o] - rm->f[o] * sc;
        for (int i = 0; i < cin; ++i)
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx) {
                    const float v = w->f[(((size_t)o * cin + i) * ks + ky) * ks + kx] * sc;
                    const size_t k = tap_major ? ((size_t)(ky * ks + kx) * cin_pad + i) : (((size_t)i * ks + ky) * ks + kx);
                    wf[(size_t)o * K + k] = v;
                }
    }
    FERN_TRY(upload(c, wf.data(), wf.size(), &out->w));
    FERN_TRY(upload(c, bf.data(), bf.size(), &out->b));
    out->cout = cout_pad;
    out->k = K;
    return FERN_OK;
}

static int finalize_resnet(fern_ctx* c, const fern_clip_config* cfg) {
    ResNetW& R = c->clip.res;
    R = ResNetW();
    const int w = cfg->r_width, half = w / 2;
    if (w <= 0 || w % 16 || cfg->image_size % 32 || cfg->r_heads <= 0) return fail(FERN_ERR_ARG, "clip: unsupported ModifiedResNet shape");
    const int embed = w * 32, hd = embed / cfg->r_heads, tokens = (cfg->image_size / 32) * (cfg->image_size / 32) + 1;
    if (embed % cfg->r_heads || hd % 4 || hd > 96 || tokens > 224) return fail(FERN_ERR_ARG, "clip: unsupported attention-pool shape");
    R.stem_c = (half + 15) / 16 * 16;
    std::vector<float> z(64, 0.f);
    FERN_TRY(upload(c, z.data(), z.size(), &R.zeros));
    FERN_TRY(up_conv_bn(c, "visual.conv1", "visual.bn1", half, 3, 3, R.stem_c, 3, false, &R.stem1));
    FERN_TRY(up_conv_bn(c, "visual.conv2", "visual.bn2", half, half, 3, R.stem_c, R.stem_c, true, &R.stem2));
    FERN_TRY(up_conv_bn(c, "visual.conv3", "visual.bn3", w, half, 3, w, R.stem_c, true, &R.stem3));
    int inplanes = w;
    for (int li = 0; li < 4; ++li) {
        const int planes = w << li;
        for (int bi = 0; bi < cfg->r_layers[li]; ++bi) {
            const std::string p = "visual.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            BottleneckW B;
            B.stride = (bi == 0 && li > 0) ? 2 : 1;
            B.cin = inplanes;
            B.planes = planes;
            FERN_TRY(up_conv_bn(c, p + ".conv1", p + ".bn1", planes, inplanes, 1, planes, inplanes, true, &B.c1));
            FERN_TRY(up_conv_bn(c, p + ".conv2", p + ".bn2", planes, planes, 3, planes, planes, true, &B.c2));
            FERN_TRY(up_conv_bn(c, p + ".conv3", p + ".bn3", planes * 4, planes, 1, planes * 4, planes, true, &B.c3));
            B.has_down = B.stride > 1 || inplanes != planes * 4;
            if (B.has_down)
                FERN_TRY(up_conv_bn(c, p + ".downsample.0", p + ".downsample.1", planes * 4, inplanes, 1, planes * 4, inplanes, true, &B.down));
            R.blocks.push_back(B);
            inplanes = planes * 4;
        }
    }
    if (inplanes != embed) return fail(FERN_ERR_ARG, "clip: ModifiedResNet needs four non-empty stages");
    FERN_TRY(up_key(c, "visual.attnpool.positional_embedding", {tokens, embed}, &R.pos));
    FERN_TRY(up_linear(c, "visual.attnpool.q_proj", embed, embed, &R.q));
    FERN_TRY(up_packed(c, {"visual.attnpool.k_proj", "visual.attnpool.v_proj"}, embed, embed, &R.kv));
    FERN_TRY(up_linear(c, "visual.attnpool.c_proj", cfg->embed_dim, embed, &R.cproj));
    return FERN_OK;
}

extern "C" int fern_finalize_clip(fern_ctx* c, const fern_clip_config* cfg) {
    if (!c || !cfg) return fail(FERN_ERR_ARG, "fern_finalize_clip: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    ClipW& W = c->clip;
    W = ClipW();
    FERN_TRY(begin_group(c, fern_ctx::G_CLIP));
    W.cfg = *cfg;
    auto bad = [](int w, int heads) { return w <= 0 || w % 32 || w > 1024 || heads <= 0 || w % heads || (w / heads) % 4 || w / heads > 96; };
    if (cfg->embed_dim <= 0 || cfg->embed_dim % 4 || cfg->embed_dim > 1024) return fail(FERN_ERR_ARG, "clip: unsupported embed_dim");
    if (cfg->t_layers > 0 && (bad(cfg->t_width, cfg->t_heads) || cfg->t_mlp % 32 || cfg->context_length > 96))
        return fail(FERN_ERR_ARG, "clip: unsupported text tower shape");
    if (cfg->v_arch == 1) {
        FERN_TRY(finalize_resnet(c, cfg));
    } else if (cfg->v_layers > 0) {
        const int g = cfg->patch_size > 0 ? cfg->image_size / cfg->patch_size : 0;
        if (bad(cfg->v_width, cfg->v_heads) || cfg->v_mlp % 32 || cfg->patch_size % 4 || g * cfg->patch_size != cfg->image_size ||
            (3 * cfg->patch_size * cfg->patch_size) % 32 || g * g + 1 > 224)
            return fail(FERN_ERR_ARG, "clip: unsupported image tower shape");
        const int vw = cfg->v_width, P = cfg->patch_size, tokens = g * g + 1;
        FERN_TRY(up_key(c, "visual.conv1.weight", {vw, 3, P, P}, &W.conv_w));
        W.conv_mx = LinearW{W.conv_w, nullptr, vw, 3 * P * P};
        FERN_TRY(make_mx8(c, &W.conv_mx));
        FERN_TRY(make_bf16(c, &W.conv_mx));
        FERN_TRY(up_key(c, "visual.class_embedding", {vw}, &W.cls));
        FERN_TRY(up_key(c, "visual.positional_embedding", {tokens, vw}, &W.vpos));
        FERN_TRY(up_ln(c, "visual.ln_pre", vw, &W.ln_pre));
        FERN_TRY(up_ln(c, "visual.ln_post", vw, &W.ln_post));
        FERN_TRY(up_transposed(c, "visual.proj", vw, cfg->embed_dim, &W.vproj_t));
        W.vblocks.resize(cfg->v_layers);
        for (int i = 0; i < cfg->v_layers; ++i)
            FERN_TRY(up_clip_block(c, "visual.transformer.resblocks." + std::to_string(i), vw, cfg->v_mlp, &W.vblocks[i]));
    }
    if (cfg->t_layers > 0) {
        const int tw = cfg->t_width;
        FERN_TRY(up_key(c, "token_embedding.weight", {cfg->vocab_size, tw}, &W.tok_emb));
        FERN_TRY(up_key(c, "positional_embedding", {cfg->context_length, tw}, &W.tpos));
        FERN_TRY(up_ln(c, "ln_final", tw, &W.ln_final));
        FERN_TRY(up_transposed(c, "text_projection", tw, cfg->embed_dim, &W.tproj_t));
        W.tblocks.resize(cfg->t_layers);
        for (int i = 0; i < cfg->t_layers; ++i)
            FERN_TRY(up_clip_block(c, "transformer.resblocks." + std::to_string(i), tw, cfg->t_mlp, &W.tblocks[i]));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));      // bf16 weight copies are converted on the null stream
    W.ready = true;
    return FERN_OK;
}

extern "C" int fern_set_precision(fern_ctx* c, int precision) {
    if (!c) return fail(FERN_ERR_ARG, "fern_set_precision: ctx is NULL");
    if (precision != FERN_PREC_FP32 && precision != FERN_PREC_BF16 && precision != FERN_PREC_FP8 && precision != FERN_PREC_MX8 &&
        precision != FERN_PREC_F32X3 && precision != FERN_PREC_MX8_MLP && precision != FERN_PREC_MX8_IMG)
        return fail(FERN_ERR_ARG, "fern_set_precision: unknown precision");
    const bool split = precision == FERN_PREC_F32X3;
    if (split) precision = FERN_PREC_FP32;         // fp32 data flow; only the GEMM arithmetic changes
    if ((precision == FERN_PREC_MX8 || precision == FERN_PREC_MX8_MLP || precision == FERN_PREC_MX8_IMG) && c->clip.ready) {
        for (const auto* blocks : {&c->clip.vblocks, &c->clip.tblocks})
            for (const auto& b : *blocks)
                if (!b.qkv.wm || !b.out.wm || !b.fc.wm || !b.proj.wm)
                    return fail(FERN_ERR_ARG, "fern_set_precision: mx8 needs tower widths and MLP widths that are multiples of 128 (<= 4096)");
    }
    if (precision == FERN_PREC_FP8 && c->clip.ready) {
        for (const auto* blocks : {&c->clip.vblocks, &c->clip.tblocks})
            for (const auto& b : *blocks)
                if (!b.qkv.w8 || !b.out.w8 || !b.fc.w8 || !b.proj.w8)
                    return fail(FERN_ERR_ARG, "fern_set_precision: fp8 needs tower widths and MLP widths that are multiples of 64 (<= 4096)");
    }
    c->precision = precision;                      // both fields are committed together, after every check has passed
    c->f32x3 = split;
    return FERN_OK;
}
extern "C" int fern_get_precision(fern_ctx* c) { return !c ? FERN_ERR_ARG : c->f32x3 ? FERN_PREC_F32X3 : c->precision; }

// ------------------------------------------------------------------------------------------------
// fusion building blocks (internal; workspace comes from the caller's arena)
// ------------------------------------------------------------------------------------------------
// VisualSR.forward (fusion_model.py:141-154): the [13n, D] local embedding is never materialised --
// the GEMM epilogue applies BN13+tanh, multiplies by the global embedding and reduces against w_common.
static int run_visual_sr(fern_ctx* c, const SRW& W, const float* local, float* out, long n, int D, hipStream_t s) {
    float *raw, *g, *partial;
    FERN_TRY(ws_get(c, (size_t)n * D, &raw));
    FERN_TRY(ws_get(c, (size_t)n * D, &g));
    HIP_TRY(launch_mean_rows(local, D, raw, D, n, 13, D, 13, 0, s));
    GemmParams pg = gemm_desc(raw, D, W.global, g, D, (int)n, EPI_COLAFFINE_TANH);
    pg.aux0 = W.bnd_scale; pg.aux1 = W.bnd_shift;
    FERN_TRY(run_gemm(c, pg, s));
    const int M = (int)(n * 13);
    const int nb = gemm_num_col_blocks(M, D, D);
    FERN_TRY(ws_get(c, (size_t)M * nb, &partial));
    GemmParams pl = gemm_desc(local, D, W.local, nullptr, D, M, EPI_SR_LOCAL);
    pl.aux0 = W.wc; pl.aux1 = W.bn13_mean; pl.aux2 = W.bn13_inv; pl.aux3 = W.bn13_beta;
    pl.G = g; pl.ldg = D; pl.partial = partial;
    FERN_TRY(run_gemm(c, pl, s));
    HIP_TRY(launch_sr_finalize(partial, nb, W.bc, local, out, n, D, s));
    return FERN_OK;
}

// CombinerSimple.forward (fusion_model.py:86-94): the 8D hidden layer only exists inside the GEMM
// epilogue, which reduces it against dynamic_scalar.3.weight on the fly.
static int run_combiner(fern_ctx* c, const CombinerW& W, const float* image, const float* text, float* out, long n, int D, hipStream_t s) {
    float *cat, *partial;
    const int Pj = W.text.out, H = 2 * Pj;      // cat(text_proj, image_proj) width; ERN: 8D
    FERN_TRY(ws_get(c, (size_t)n * H, &cat));
    FERN_TRY(run_gemm(c, gemm_desc(text, D, W.text, cat, H, (int)n, EPI_BIAS_RELU), s));          // :87,:90 text first
    FERN_TRY(run_gemm(c, gemm_desc(image, D, W.image, cat + Pj, H, (int)n, EPI_BIAS_RELU), s));   // :88
    const int nb = gemm_num_col_blocks((int)n, W.hidden.out, H);
    FERN_TRY(ws_get(c, (size_t)n * nb, &partial));
    GemmParams ph = gemm_desc(cat, H, W.hidden, nullptr, H, (int)n, EPI_RELU_DOT);
    ph.aux0 = W.w2; ph.partial = partial;
    if (W.ksplit > 1) {
        float* kpart;
        FERN_TRY(ws_get(c, (size_t)W.ksplit * n * W.hidden.out, &kpart));
        ph.ksplit = W.ksplit; ph.kpart = kpart; ph.epi = EPI_BIAS;      // slices store raw sums; bias + ReLU + dot happen in the reduce
        FERN_TRY(run_gemm(c, ph, s));
        HIP_TRY(launch_splitk_relu_dot(kpart, W.ksplit, n, W.hidden.out, W.hidden.b, W.w2, partial, s));
    } else {
        FERN_TRY(run_gemm(c, ph, s));
    }
    HIP_TRY(launch_combiner_finalize(partial, nb, W.b2, image, text, out, n, D, s));
    return FERN_OK;
}

static int check_fusion(fern_ctx* c, const char* fn, int parts) {
    if (!c) return fail(FERN_ERR_ARG, std::string(fn) + ": ctx is NULL");
    if ((c->fusion.parts & parts) != parts) return fail(FERN_ERR_STATE, std::string(fn) + ": fusion weights not finalised (fern_finalize_fusion)");
    FERN_TRY(check_fresh(c, fn));
    HIP_TRY(hipSetDevice(c->device));
    return FERN_OK;
}

extern "C" int fern_combiner(fern_ctx* c, int which, const float* image, const float* text, float* out, int64_t n, void* stream) {
    if (which < 0 || which > 3) return fail(FERN_ERR_ARG, "fern_combiner: bad combiner id");
    FERN_TRY(check_fusion(c, "fern_combiner", which == FERN_COMBINER_TARGET ? FERN_PART_TARGET_COMBINER : FERN_PART_DVR));
    if ( n < 0 || (n && (!image || !text || !out))) return fail(FERN_ERR_ARG, "fern_combiner: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int D = c->fusion.D;
    const long CH = 8192;
    for (long o = 0; o < n; o += CH) {
        const long m = std::min(CH, (long)n - o);
        FERN_TRY(ws_begin(c, s));
        FERN_TRY(run_combiner(c, c->fusion.comb[which], image + o * D, text + o * D, out + o * D, m, D, s));
    }
    return FERN_OK;
}

extern "C" int fern_visual_sr(fern_ctx* c, int which, const float* local, float* out, int64_t n, void* stream) {
    if (which < 0 || which > 1) return fail(FERN_ERR_ARG, "fern_visual_sr: bad VisualSR id");
    FERN_TRY(check_fusion(c, "fern_visual_sr", which == FERN_SR_TARGET ? FERN_PART_TARGET_SR : FERN_PART_DVR));
    if ( n < 0 || (n && (!local || !out))) return fail(FERN_ERR_ARG, "fern_visual_sr: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int D = c->fusion.D;
    const long CH = 8192;
    for (long o = 0; o < n; o += CH) {
        const long m = std::min(CH, (long)n - o);
        FERN_TRY(ws_begin(c, s));
        FERN_TRY(run_visual_sr(c, c->fusion.sr[which], local + o * 13 * D, out + o * D, m, D, s));
    }
    return FERN_OK;
}

// models/others/Combiner_Model.py (CLIP4Cir Combiner, CVPR'22): the variant of CombinerSimple with a residual
// output_layer(relu(combiner_layer(cat))) branch; its Linear layers take inputs of width 2 * clip_feature_dim.
extern "C" int fern_finalize_clip4cir(fern_ctx* c) {
    if (!c) return fail(FERN_ERR_ARG, "fern_finalize_clip4cir: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    Clip4CirW& W = c->c4c;
    W = Clip4CirW();
    FERN_TRY(begin_group(c, fern_ctx::G_CLIP4CIR));
    const std::string p = "clip4cir.";
    const HostTensor *tp, *cl;
    FERN_TRY(need(c, p + "text_projection_layer.weight", {}, &tp));
    FERN_TRY(need(c, p + "combiner_layer.weight", {}, &cl));
    if (tp->shape.size() != 2 || cl->shape.size() != 2) return fail(FERN_ERR_STATE, "clip4cir: weights must be 2-D");
    const int Pj = (int)tp->shape[0], Wd = (int)tp->shape[1], Hd = (int)cl->shape[0];
    if (Wd % 32 || Wd > 1280 || Pj % 16 || Hd % 32) return fail(FERN_ERR_ARG, "clip4cir: need 2*clip_dim % 32 == 0 (<= 1280), projection % 16, hidden % 32");
    FERN_TRY(up_linear(c, p + "text_projection_layer", Pj, Wd, &W.text));
    FERN_TRY(up_linear(c, p + "image_projection_layer", Pj, Wd, &W.image));
    FERN_TRY(up_linear(c, p + "combiner_layer", Hd, 2 * Pj, &W.comb));
    FERN_TRY(up_linear(c, p + "output_layer", Wd, Hd, &W.outl));
    FERN_TRY(up_linear(c, p + "dynamic_scalar.0", Hd, 2 * Pj, &W.hidden));
    FERN_TRY(up_key(c, p + "dynamic_scalar.3.weight", {1, Hd}, &W.w2));
    FERN_TRY(up_key(c, p + "dynamic_scalar.3.bias", {1}, &W.b2));
    W.width = Wd;
    W.ready = true;
    return FERN_OK;
}

extern "C" int fern_combiner_clip4cir(fern_ctx* c, const float* image, const float* text, float* out, int64_t n, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_combiner_clip4cir: ctx is NULL");
    if (!c->c4c.ready) return fail(FERN_ERR_STATE, "fern_combiner_clip4cir: weights not finalised (fern_finalize_clip4cir)");
    FERN_TRY(check_fresh(c, "fern_combiner_clip4cir"));
    if (n < 0 || (n && (!image || !text || !out))) return fail(FERN_ERR_ARG, "fern_combiner_clip4cir: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const Clip4CirW& W = c->c4c;
    const int Wd = W.width, Pj = W.text.out, H2 = 2 * Pj, Hd = W.comb.out;
    const long CH = 8192;
    for (long o = 0; o < n; o += CH) {
        const int m = (int)std::min(CH, (long)n - o);
        FERN_TRY(ws_begin(c, s));
        float *cat, *comb, *O, *partial;
        FERN_TRY(ws_get(c, (size_t)m * H2, &cat));
        FERN_TRY(ws_get(c, (size_t)m * Hd, &comb));
        FERN_TRY(ws_get(c, (size_t)m * Wd, &O));
        const float *im = image + o * Wd, *tx = text + o * Wd;
        FERN_TRY(run_gemm(c, gemm_desc(tx, Wd, W.text, cat, H2, m, EPI_BIAS_RELU), s));              // :49-51
        FERN_TRY(run_gemm(c, gemm_desc(im, Wd, W.image, cat + Pj, H2, m, EPI_BIAS_RELU), s));        // :52-54
        FERN_TRY(run_gemm(c, gemm_desc(cat, H2, W.comb, comb, Hd, m, EPI_BIAS_RELU), s));            // :59-61
        FERN_TRY(run_gemm(c, gemm_desc(comb, Hd, W.outl, O, Wd, m, EPI_BIAS), s));                   // output_layer
        const int nb = gemm_num_col_blocks(m, Hd, H2);
        FERN_TRY(ws_get(c, (size_t)m * nb, &partial));
        GemmParams ph = gemm_desc(cat, H2, W.hidden, nullptr, H2, m, EPI_RELU_DOT);                  // dynamic_scalar
        ph.aux0 = W.w2; ph.partial = partial;
        FERN_TRY(run_gemm(c, ph, s));
        HIP_TRY(launch_combiner_finalize(partial, nb, W.b2, im, tx, out + o * Wd, m, Wd, s, O));     // :63-70
    }
    return FERN_OK;
}

// utils.element_wise_sum (utils/utils.py:133-140): F.normalize(image_features + text_features)
extern "C" int fern_element_wise_sum(fern_ctx* c, const float* image, const float* text, float* out, int64_t n, int d, void* stream) {
    if (!c || n < 0 || (n && (!image || !text || !out))) return fail(FERN_ERR_ARG, "fern_element_wise_sum: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_l2norm(image, d, out, d, n, d, 1e-12f, 0, (hipStream_t)stream, text));
    return FERN_OK;
}

// losses/loss.py:10-14 BatchBasedClassificationLoss.forward: cross_entropy(100 * predicted @ target.T, arange(B)), forward value
extern "C" int fern_batch_classification_loss(fern_ctx* c, const float* predicted, const float* target, int B, int D, float* out_loss,
                                              void* stream) {
    if (!c || B < 1 || D < 1 || !predicted || !target || !out_loss) return fail(FERN_ERR_ARG, "fern_batch_classification_loss: bad argument");
    if (D % 16) return fail(FERN_ERR_ARG, "fern_batch_classification_loss: D must be a multiple of 16");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    FERN_TRY(ws_begin(c, s));
    const long ld = (B + 3) & ~3L;
    float *logits, *rows;
    FERN_TRY(ws_get(c, (size_t)B * ld, &logits));
    FERN_TRY(ws_get(c, (size_t)B, &rows));
    GemmParams p{};
    p.A = predicted; p.lda = D; p.W = target; p.ldw = D; p.C = logits; p.ldc = ld; p.M = B; p.N = B; p.K = D; p.epi = EPI_BIAS; p.aload = ALOAD_PLAIN;
    FERN_TRY(run_gemm(c, p, s));
    HIP_TRY(launch_ce_diag_mean(logits, ld, B, 100.0f, rows, out_loss, s));
    return FERN_OK;
}

extern "C" int fern_l2_normalize(fern_ctx* c, const float* x, float* out, int64_t n, int d, void* stream) {
    if (!c || n < 0 || (n && (!x || !out))) return fail(FERN_ERR_ARG, "fern_l2_normalize: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_l2norm(x, d, out, d, n, d, 1e-12f, 0, (hipStream_t)stream));
    return FERN_OK;
}

extern "C" int fern_index_fuse(fern_ctx* c, const float* tar_feats, const float* tar_local, float* out, int64_t n, int normalize_input,
                               void* stream) {
    FERN_TRY(check_fusion(c, "fern_index_fuse", FERN_PART_TARGET_SR | FERN_PART_TARGET_COMBINER));
    if (n < 0 || (n && (!tar_feats || !tar_local || !out))) return fail(FERN_ERR_ARG, "fern_index_fuse: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int D = c->fusion.D;
    const long CH = 8192;   // gallery rows per tile: bounds the [rows, 8D] Combiner buffer (the reference runs the whole gallery as ONE batch)
    for (long o = 0; o < n; o += CH) {
        const long m = std::min(CH, (long)n - o);
        FERN_TRY(ws_begin(c, s));
        const float* img = tar_feats + o * D;
        if (normalize_input) {
            float* tn;
            FERN_TRY(ws_get(c, (size_t)m * D, &tn));
            HIP_TRY(launch_l2norm(img, D, tn, D, m, D, 1e-12f, 0, s));      // F.normalize, test_fiq.py:45
            img = tn;
        }
        float* sr;
        FERN_TRY(ws_get(c, (size_t)m * D, &sr));
        FERN_TRY(run_visual_sr(c, c->fusion.sr[FERN_SR_TARGET], tar_local + o * 13 * D, sr, m, D, s));      // model.py:65
        FERN_TRY(run_combiner(c, c->fusion.comb[FERN_COMBINER_TARGET], img, sr, out + o * D, m, D, s));     // model.py:66
    }
    return FERN_OK;
}

// ERN mode="test" (model.py:68-69 -> DVR_module.forward, fusion_model.py:26-55)
static int dvr_chunk(fern_ctx* c, const float* ref_global, const float* ref_local, const float* text_global, const float* text_seq,
                     float* out, int B, int T, hipStream_t s) {
    const FusionW& F = c->fusion;
    const int D = F.D, P = 13, S = 1 + P + T, heads = 8, hd = D / heads;
    const long R = (long)B * S;
    const int inter = F.layer[0].inter.out;
    float *X, *X1, *QKV, *ATT, *H;
    FERN_TRY(ws_get(c, (size_t)R * D, &X));
    FERN_TRY(ws_get(c, (size_t)R * D, &X1));
    FERN_TRY(ws_get(c, (size_t)R * 3 * D, &QKV));
    FERN_TRY(ws_get(c, (size_t)R * D, &ATT));
    FERN_TRY(ws_get(c, (size_t)R * inter, &H));
    // PlusModel / BertEmbeddings (fusion_model.py:199-212)
    HIP_TRY(launch_bert_embed(F.cls, ref_local, text_seq, F.type, F.pos, F.emb_ln.g, F.emb_ln.b, X, B, P, T, D, 1e-12f, s));
    const bool reduced = c->precision != FERN_PREC_FP32;
    unsigned short* Xb = nullptr;
    if (reduced) FERN_TRY(ws_get(c, (size_t)R * D, &Xb));
    for (int l = 0; l < 2; ++l) {
        const BertLayerW& L = F.layer[l];
        if (reduced) {
            // Reduced-precision modes (bf16 and fp8 alike): the two BERT blocks follow the towers' bf16 recipe -- bf16 operands
            // on the four token-level GEMMs and the attention, fp32 accumulation, fp32 residual stream and LayerNorm.
            unsigned short* QKVb = reinterpret_cast<unsigned short*>(QKV);
            unsigned short* ATTb = reinterpret_cast<unsigned short*>(ATT);
            unsigned short* Hb = reinterpret_cast<unsigned short*>(H);
            HIP_TRY(launch_f32_to_bf16(X, Xb, R * D, s));
            FERN_TRY(run_gemm_b(c, gemm_desc_b(Xb, D, L.qkv, QKVb, 3 * D, (int)R, EPI_BIAS, true), s));
            AttnParams ab{nullptr, nullptr, nullptr, nullptr, 3L * D, 3L * D, 3L * D, (long)D, B, heads, hd, S, S, 0,
                          1.0f / std::sqrt((float)hd), ATTb, QKVb, QKVb + D, QKVb + 2 * D};
            FERN_TRY(run_attention(c, ab, s));
            GemmParams qo = gemm_desc_b(ATTb, D, L.attn_out, X1, D, (int)R, EPI_BIAS_RESIDUAL, false);
            qo.R = X;
            FERN_TRY(run_gemm_b(c, qo, s));
            HIP_TRY(launch_layernorm(X1, nullptr, L.ln1.g, L.ln1.b, X1, R, D, D, D, 1e-12f, s));
            HIP_TRY(launch_f32_to_bf16(X1, Xb, R * D, s));
            FERN_TRY(run_gemm_b(c, gemm_desc_b(Xb, D, L.inter, Hb, inter, (int)R, EPI_BIAS_GELU, true), s));
            GemmParams q2 = gemm_desc_b(Hb, inter, L.out, X, D, (int)R, EPI_BIAS_RESIDUAL, false);
            q2.R = X1;
            FERN_TRY(run_gemm_b(c, q2, s));
            HIP_TRY(launch_layernorm(X, nullptr, L.ln2.g, L.ln2.b, X, R, D, D, D, 1e-12f, s));
            continue;
        }
        FERN_TRY(run_gemm(c, gemm_desc(X, D, L.qkv, QKV, 3 * D, (int)R, EPI_BIAS), s));
        AttnParams a{QKV, QKV + D, QKV + 2 * D, ATT, 3L * D, 3L * D, 3L * D, (long)D, B, heads, hd, S, S, 0, 1.0f / std::sqrt((float)hd)};
        FERN_TRY(run_attention(c, a, s));
        GemmParams po = gemm_desc(ATT, D, L.attn_out, X1, D, (int)R, EPI_BIAS_RESIDUAL);
        po.R = X;
        FERN_TRY(run_gemm(c, po, s));
        HIP_TRY(launch_layernorm(X1, nullptr, L.ln1.g, L.ln1.b, X1, R, D, D, D, 1e-12f, s));
        FERN_TRY(run_gemm(c, gemm_desc(X1, D, L.inter, H, inter, (int)R, EPI_BIAS_GELU), s));
        GemmParams p2 = gemm_desc(H, inter, L.out, X, D, (int)R, EPI_BIAS_RESIDUAL);
        p2.R = X1;
        FERN_TRY(run_gemm(c, p2, s));
        HIP_TRY(launch_layernorm(X, nullptr, L.ln2.g, L.ln2.b, X, R, D, D, D, 1e-12f, s));
    }
    // F.normalize every row of last_hidden_state (rows 1..13 image, 14.. text; row 0 is never read) -- :38-41
    float* XN = X1;
    HIP_TRY(launch_l2norm(X, D, XN, D, R, D, 1e-12f, 0, s));
    // MR_component: only output rows 0..12 survive (:47), so Q is projected for the first 13 text rows only
    const long Rp = (long)B * P;
    float *IMG, *TXT, *Q13, *KV, *CA, *CROSS, *PV, *TM, *G, *Lf;
    FERN_TRY(ws_get(c, (size_t)Rp * D, &IMG));
    FERN_TRY(ws_get(c, (size_t)Rp * D, &TXT));
    FERN_TRY(ws_get(c, (size_t)Rp * D, &Q13));
    FERN_TRY(ws_get(c, (size_t)Rp * 2 * D, &KV));
    FERN_TRY(ws_get(c, (size_t)Rp * D, &CA));
    FERN_TRY(ws_get(c, (size_t)Rp * D, &CROSS));
    FERN_TRY(ws_get(c, (size_t)B * D, &PV));
    FERN_TRY(ws_get(c, (size_t)B * D, &TM));
    FERN_TRY(ws_get(c, (size_t)B * D, &G));
    FERN_TRY(ws_get(c, (size_t)B * D, &Lf));
    HIP_TRY(launch_gather_rows(XN, D, IMG, D, Rp, D, P, S, 1, nullptr, s));
    HIP_TRY(launch_gather_rows(XN, D, TXT, D, Rp, D, P, S, 1 + P, nullptr, s));
    FERN_TRY(run_gemm(c, gemm_desc(TXT, D, F.mha_q, Q13, D, (int)Rp, EPI_BIAS), s));
    FERN_TRY(run_gemm(c, gemm_desc(IMG, D, F.mha_kv, KV, 2 * D, (int)Rp, EPI_BIAS), s));
    AttnParams ca{Q13, KV, KV + D, CA, (long)D, 2L * D, 2L * D, (long)D, B, heads, hd, P, P, 0, 1.0f / std::sqrt((float)hd)};
    FERN_TRY(run_attention(c, ca, s));
    FERN_TRY(run_gemm(c, gemm_desc(CA, D, F.mha_out, CROSS, D, (int)Rp, EPI_BIAS), s));
    FERN_TRY(run_visual_sr(c, F.sr[FERN_SR_DVR], CROSS, PV, B, D, s));                                   // :48
    HIP_TRY(launch_mean_rows(XN, D, TM, D, B, T, D, S, 1 + P, s));                                       // :49
    FERN_TRY(run_combiner(c, F.comb[FERN_COMBINER_DVR_GLOBAL], ref_global, text_global, G, B, D, s));    // :52
    FERN_TRY(run_combiner(c, F.comb[FERN_COMBINER_DVR_LOCAL], PV, TM, Lf, B, D, s));                     // :53
    return run_combiner(c, F.comb[FERN_COMBINER_DVR_FINAL], G, Lf, out, B, D, s);                        // :54
}

extern "C" int fern_dvr_fuse(fern_ctx* c, const float* ref_global, const float* ref_local, const float* text_global, const float* text_seq,
                             float* out, int B, int seq_len, void* stream) {
    FERN_TRY(check_fusion(c, "fern_dvr_fuse", FERN_PART_DVR));
    if (B < 0 || (B && (!ref_global || !ref_local || !text_global || !text_seq || !out))) return fail(FERN_ERR_ARG, "fern_dvr_fuse: bad argument");
    // the MR cross-attention keeps output rows [:13] of the text queries (fusion_model.py:47) and feeds them to BatchNorm1d(13):
    // fewer than 13 text rows fail there in the reference; here they would read the next sample's rows
    if (seq_len < 13 || 1 + 13 + seq_len > 96) return fail(FERN_ERR_ARG, "fern_dvr_fuse: need 13 <= seq_len and 1 + 13 + seq_len <= 96");
    hipStream_t s = (hipStream_t)stream;
    const int D = c->fusion.D, CH = 256;
    for (int o = 0; o < B; o += CH) {
        const int m = std::min(CH, B - o);
        FERN_TRY(ws_begin(c, s));
        FERN_TRY(dvr_chunk(c, ref_global + (long)o * D, ref_local + (long)o * 13 * D, text_global + (long)o * D,
                           text_seq + (long)o * seq_len * D, out + (long)o * D, m, seq_len, s));
    }
    return FERN_OK;
}

// ------------------------------------------------------------------------------------------------
// CLIP towers
// ------------------------------------------------------------------------------------------------
// pre-LN residual block (modeling_clip.py:354-401): x += attn(ln_1(x)); x += mlp(ln_2(x))
static int clip_block(fern_ctx* c, const ClipBlockW& Bk, float* X, float* XN, float* QKV, float* ATT, float* H, int batch, int S,
                      int width, int heads, int causal, hipStream_t s) {
    const long R = (long)batch * S;
    const int hd = width / heads;
    HIP_TRY(launch_layernorm(X, nullptr, Bk.ln1.g, Bk.ln1.b, XN, R, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm(c, gemm_desc(XN, width, Bk.qkv, QKV, 3 * width, (int)R, EPI_BIAS), s));
    AttnParams a{QKV, QKV + width, QKV + 2 * width, ATT, 3L * width, 3L * width, 3L * width, (long)width,
                 batch, heads, hd, S, S, causal, 1.0f / std::sqrt((float)hd)};
    FERN_TRY(run_attention(c, a, s));
    GemmParams po = gemm_desc(ATT, width, Bk.out, X, width, (int)R, EPI_BIAS_RESIDUAL);
    po.R = X;
    FERN_TRY(run_gemm(c, po, s));
    HIP_TRY(launch_layernorm(X, nullptr, Bk.ln2.g, Bk.ln2.b, XN, R, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm(c, gemm_desc(XN, width, Bk.fc, H, Bk.fc.out, (int)R, EPI_BIAS_GELU), s));
    GemmParams p2 = gemm_desc(H, Bk.fc.out, Bk.proj, X, width, (int)R, EPI_BIAS_RESIDUAL);
    p2.R = X;
    return run_gemm(c, p2, s);
}

// Perf-mode block: the four token-level GEMMs take bf16 operands (LayerNorm / attention / GELU outputs are rounded to bf16
// as they are written, weights are the bf16 copies) and accumulate in fp32; the residual stream X, the LayerNorm
// statistics and the attention softmax stay fp32.
static int clip_block_bf16(fern_ctx* c, const ClipBlockW& Bk, float* X, unsigned short* XNb, unsigned short* QKVb, unsigned short* ATTb,
                           unsigned short* Hb, int batch, int S, int width, int heads, int causal, hipStream_t s) {
    const long R = (long)batch * S;
    const int hd = width / heads;
    HIP_TRY(launch_layernorm_bf16(X, Bk.ln1.g, Bk.ln1.b, XNb, R, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm_b(c, gemm_desc_b(XNb, width, Bk.qkv, QKVb, 3 * width, (int)R, EPI_BIAS, true), s));
    AttnParams a{nullptr, nullptr, nullptr, nullptr, 3L * width, 3L * width, 3L * width, (long)width,
                 batch, heads, hd, S, S, causal, 1.0f / std::sqrt((float)hd), ATTb, QKVb, QKVb + width, QKVb + 2 * width};
    FERN_TRY(run_attention(c, a, s));
    GemmParams po = gemm_desc_b(ATTb, width, Bk.out, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
    po.R = X;
    FERN_TRY(run_gemm_b(c, po, s));
    HIP_TRY(launch_layernorm_bf16(X, Bk.ln2.g, Bk.ln2.b, XNb, R, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm_b(c, gemm_desc_b(XNb, width, Bk.fc, Hb, Bk.fc.out, (int)R, EPI_BIAS_GELU, true), s));
    GemmParams p2 = gemm_desc_b(Hb, Bk.fc.out, Bk.proj, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
    p2.R = X;
    return run_gemm_b(c, p2, s);
}

// fp8 block (BASELINE config 5: "fp8 MFMA encoder path"): the four token-level GEMMs take e4m3fn operands with per-token
// activation scales and per-output-channel weight scales, folded back in the fp32 epilogue.  LayerNorm quantises as it
// writes; the attention and GELU outputs are written as bf16 (a row's maximum is not known inside those kernels) and
// quantised by a row pass.  Attention itself runs in the bf16 operand form; everything else as in the bf16 block.
static int clip_block_fp8(fern_ctx* c, const ClipBlockW& Bk, float* X, unsigned char* XN8, float* SA, unsigned short* QKVb,
                          unsigned short* ATTb, unsigned char* ATT8, unsigned short* Hb, unsigned char* H8, int batch, int S, int width,
                          int heads, int causal, hipStream_t s) {
    const long R = (long)batch * S;
    const int hd = width / heads, mlp = Bk.fc.out;
    HIP_TRY(launch_layernorm_fp8(X, Bk.ln1.g, Bk.ln1.b, XN8, SA, R, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm_b(c, gemm_desc_f8(XN8, SA, width, Bk.qkv, QKVb, 3 * width, (int)R, EPI_BIAS, true), s));
    AttnParams a{nullptr, nullptr, nullptr, nullptr, 3L * width, 3L * width, 3L * width, (long)width,
                 batch, heads, hd, S, S, causal, 1.0f / std::sqrt((float)hd), ATTb, QKVb, QKVb + width, QKVb + 2 * width};
    FERN_TRY(run_attention(c, a, s));
    HIP_TRY(launch_quantize_rows_fp8(ATTb, nullptr, width, ATT8, width, SA, R, width, s));
    GemmParams po = gemm_desc_f8(ATT8, SA, width, Bk.out, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
    po.R = X;
    FERN_TRY(run_gemm_b(c, po, s));
    HIP_TRY(launch_layernorm_fp8(X, Bk.ln2.g, Bk.ln2.b, XN8, SA, R, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm_b(c, gemm_desc_f8(XN8, SA, width, Bk.fc, Hb, mlp, (int)R, EPI_BIAS_GELU, true), s));
    HIP_TRY(launch_quantize_rows_fp8(Hb, nullptr, mlp, H8, mlp, SA, R, mlp, s));
    GemmParams p2 = gemm_desc_f8(H8, SA, mlp, Bk.proj, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
    p2.R = X;
    return run_gemm_b(c, p2, s);
}

// MX block (FERN_PREC_MX8): clip_block_fp8 with block-scaled operands -- one E8M0 scale per (token, 32 channels) and per (output
// channel, 32 inputs), applied inside v_mfma_scale_f32_32x32x64_f8f6f4 (twice the MFMA rate of the plain fp8 form, and an outlier
// channel costs its own 32-block precision, not the whole token row).  SM: the activation scales of the operand in flight.
// The residual stream of this mode is BF16 (Xb, round 4): the two residual GEMMs of a block read and write it in place
// (C = bf16(acc + bias + Xb): one rounding per residual add) and the LayerNorms read it -- out-proj / c_proj were HBM-bound on their
// 2 x 39 MB of fp32 residual traffic per launch (M = 12608), the LayerNorm + quantise passes on their 39 MB reads.
static int clip_block_mx8(fern_ctx* c, const ClipBlockW& Bk, unsigned short* Xb, unsigned char* XN8, unsigned char* SM, unsigned short* QKVb,
                          unsigned short* ATTb, unsigned char* ATT8, unsigned char* H8, int batch, int S, int width,
                          int heads, int causal, hipStream_t s) {
    const long R = (long)batch * S;
    const int hd = width / heads, mlp = Bk.fc.out;
    HIP_TRY(launch_layernorm_mx8(nullptr, Bk.ln1.g, Bk.ln1.b, XN8, SM, R, R, width, width, width, 1e-5f, s, Xb));
    FERN_TRY(run_gemm_b(c, gemm_desc_mx(XN8, SM, R, width, Bk.qkv, QKVb, 3 * width, (int)R, EPI_BIAS, true), s));
    AttnParams a{nullptr, nullptr, nullptr, nullptr, 3L * width, 3L * width, 3L * width, (long)width,
                 batch, heads, hd, S, S, causal, 1.0f / std::sqrt((float)hd), ATTb, QKVb, QKVb + width, QKVb + 2 * width};
    if (hd % 32 == 0) {      // the attention kernel quantises its fp32 output itself (the QKV GEMM is done with SM by now)
        a.out_b = nullptr; a.out_q8 = ATT8; a.out_scales = SM; a.out_srows = R;
        FERN_TRY(run_attention(c, a, s));
    } else {
        FERN_TRY(run_attention(c, a, s));
        HIP_TRY(launch_quantize_mx8(ATTb, nullptr, width, ATT8, width, SM, R, R, width, s));
    }
    GemmParams po = gemm_desc_mx(ATT8, SM, R, width, Bk.out, Xb, width, (int)R, EPI_BIAS_RESIDUAL, true);
    po.Rb = Xb;
    FERN_TRY(run_gemm_b(c, po, s));
    HIP_TRY(launch_layernorm_mx8(nullptr, Bk.ln2.g, Bk.ln2.b, XN8, SM, R, R, width, width, width, 1e-5f, s, Xb));
    // c_fc quantises its GELU output where it is produced (fp32 values -> e4m3fn + block scales): no bf16 round trip, no extra pass
    unsigned char* SH = SM + ((size_t)R * (width / 32) + 255) / 256 * 256;      // H's scales, behind the LayerNorm output's
    GemmParams pf = gemm_desc_mx(XN8, SM, R, width, Bk.fc, H8, mlp, (int)R, EPI_BIAS_GELU, false);
    pf.out_mx8 = 1; pf.mxc = SH; pf.mxc_rows = R;
    FERN_TRY(run_gemm_b(c, pf, s));
    GemmParams p2 = gemm_desc_mx(H8, SH, R, mlp, Bk.proj, Xb, width, (int)R, EPI_BIAS_RESIDUAL, true);
    p2.Rb = Xb;
    return run_gemm_b(c, p2, s);
}

// Mixed block (FERN_PREC_MX8_MLP): the attention half as in the bf16 block (LayerNorm -> bf16, QKV and out-proj on the bf16 MFMA, bf16
// attention), the MLP half -- two thirds of a block's GEMM flops -- as in the block-scaled block (LayerNorm -> e4m3fn + E8M0 scales,
// c_fc quantising its GELU output where it is produced, c_proj), over the fp32 residual stream of the bf16 / fp8 modes.  Half of the
// fp8 rounding points of the MX8 mode (and its bf16 residual stream) are gone; bench.py's `reduced_modes` table has what that buys.
// attn_mx (FERN_PREC_MX8_IMG, round 6): the attention half block-scaled too -- LayerNorm-1 writes e4m3fn + scales, the attention kernel
// quantises its own output, QKV and out-proj run on the scaled MFMA -- still over the FP32 residual stream: all four GEMMs at the fp8
// rate without FERN_PREC_MX8's bf16 stream, whose 24 extra roundings per tower cost more Recall than the four GEMMs' operands do.
static int clip_block_mxmlp(fern_ctx* c, const ClipBlockW& Bk, float* X, float* XN, unsigned short* QKVb, unsigned short* ATTb, unsigned char* H8,
                            int batch, int S, int width, int heads, int causal, hipStream_t s, bool attn_mx = false) {
    const long R = (long)batch * S;
    const int hd = width / heads, mlp = Bk.fc.out;
    unsigned short* XNb = reinterpret_cast<unsigned short*>(XN);
    unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
    unsigned char* SM = reinterpret_cast<unsigned char*>(XN + ((size_t)R * width / 4 + 63) / 64 * 64);
    const bool qkv_mx = attn_mx && hd % 32 == 0;
    if (qkv_mx) {
        HIP_TRY(launch_layernorm_mx8(X, Bk.ln1.g, Bk.ln1.b, XN8, SM, R, R, width, width, width, 1e-5f, s));
        FERN_TRY(run_gemm_b(c, gemm_desc_mx(XN8, SM, R, width, Bk.qkv, QKVb, 3 * width, (int)R, EPI_BIAS, true), s));
    } else {
        HIP_TRY(launch_layernorm_bf16(X, Bk.ln1.g, Bk.ln1.b, XNb, R, width, width, width, 1e-5f, s));
        FERN_TRY(run_gemm_b(c, gemm_desc_b(XNb, width, Bk.qkv, QKVb, 3 * width, (int)R, EPI_BIAS, true), s));
    }
    AttnParams a{nullptr, nullptr, nullptr, nullptr, 3L * width, 3L * width, 3L * width, (long)width,
                 batch, heads, hd, S, S, causal, 1.0f / std::sqrt((float)hd), ATTb, QKVb, QKVb + width, QKVb + 2 * width};
    if (qkv_mx) {      // the attention kernel quantises its output itself (the QKV GEMM is done with SM by now)
        unsigned char* ATT8 = reinterpret_cast<unsigned char*>(ATTb) + (size_t)R * width * 2;
        a.out_b = nullptr; a.out_q8 = ATT8; a.out_scales = SM; a.out_srows = R;
        FERN_TRY(run_attention(c, a, s));
        GemmParams po = gemm_desc_mx(ATT8, SM, R, width, Bk.out, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
        po.R = X;
        FERN_TRY(run_gemm_b(c, po, s));
    } else {
        FERN_TRY(run_attention(c, a, s));
        GemmParams po = gemm_desc_b(ATTb, width, Bk.out, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
        po.R = X;
        FERN_TRY(run_gemm_b(c, po, s));
    }
    HIP_TRY(launch_layernorm_mx8(X, Bk.ln2.g, Bk.ln2.b, XN8, SM, R, R, width, width, width, 1e-5f, s));
    unsigned char* SH = SM + ((size_t)R * (width / 32) + 255) / 256 * 256;      // H's scales, behind the LayerNorm output's
    GemmParams pf = gemm_desc_mx(XN8, SM, R, width, Bk.fc, H8, mlp, (int)R, EPI_BIAS_GELU, false);
    pf.out_mx8 = 1; pf.mxc = SH; pf.mxc_rows = R;
    FERN_TRY(run_gemm_b(c, pf, s));
    GemmParams p2 = gemm_desc_mx(H8, SH, R, mlp, Bk.proj, X, width, (int)R, EPI_BIAS_RESIDUAL, false);
    p2.R = X;
    return run_gemm_b(c, p2, s);
}

// Last ViT block: only the class token is consumed afterwards (ln_post on token 0, modeling_clip.py:876-877), so
// K/V are projected for every token but Q, the attention output, out_proj and the MLP run for the class rows only.
// Equal to the full block on the rows that are read only to TOLERANCE since round 5: the single-query attention kernel and the
// 512-wide split-K c_proj add their products in another order than the full block's kernels (fp32 parity mode included).
// Xb (FERN_PREC_MX8 with its bf16 residual stream): the stream itself -- the token-level LayerNorm reads it as the full blocks do and
// only the class rows are widened to fp32 (X is then not read: round 5 dropped the 58 MB bf16 -> fp32 pass over the whole stream).
static int clip_block_cls_only(fern_ctx* c, const ClipBlockW& Bk, const float* X, float* XN, float* QKV, float* CLS /*[b,width] out*/,
                               float* T0 /*[b,width]*/, float* T1 /*[b,width]*/, float* H /*[b,mlp]*/, int batch, int S, int width,
                               int heads, hipStream_t s, const unsigned short* Xb = nullptr) {
    const long R = (long)batch * S;
    const int hd = width / heads;
    LinearW kv{Bk.qkv.w + (size_t)width * width, Bk.qkv.b + width, 2 * width, width, Bk.qkv.wb + (size_t)width * width,
               Bk.qkv.w8 ? Bk.qkv.w8 + (size_t)width * width : nullptr, Bk.qkv.sw ? Bk.qkv.sw + width : nullptr,
               Bk.qkv.wm ? Bk.qkv.wm + (size_t)width * width : nullptr, Bk.qkv.swm ? Bk.qkv.swm + (size_t)width * 4 : nullptr, Bk.qkv.swm_rows};
    if (c->precision == FERN_PREC_MX8) {
        // MX mode: block-scaled K/V projection of all tokens (fp32 output); the class-row chain below stays fp32
        unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
        unsigned char* SM = reinterpret_cast<unsigned char*>(XN + ((size_t)R * width / 4 + 63) / 64 * 64);
        if (Xb) HIP_TRY(launch_layernorm_mx8(nullptr, Bk.ln1.g, Bk.ln1.b, XN8, SM, R, R, width, width, width, 1e-5f, s, Xb));
        else HIP_TRY(launch_layernorm_mx8(X, Bk.ln1.g, Bk.ln1.b, XN8, SM, R, R, width, width, width, 1e-5f, s));
        FERN_TRY(run_gemm_b(c, gemm_desc_mx(XN8, SM, R, width, kv, QKV + width, 3 * width, (int)R, EPI_BIAS, false), s));
        if (Xb) HIP_TRY(launch_gather_rows_bf16(Xb, width, T1, width, batch, width, S, s));
        else HIP_TRY(launch_gather_rows(X, width, T1, width, batch, width, 1, S, 0, nullptr, s));
        HIP_TRY(launch_layernorm(T1, nullptr, Bk.ln1.g, Bk.ln1.b, T0, batch, width, width, width, 1e-5f, s));
    } else if (c->precision == FERN_PREC_FP8) {
        // fp8 mode: quantised K/V projection of all tokens (fp32 output); the class-row chain below stays fp32
        unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
        float* SA = XN + ((size_t)R * width / 4 + 63) / 64 * 64;       // scales live behind the fp8 rows inside XN
        HIP_TRY(launch_layernorm_fp8(X, Bk.ln1.g, Bk.ln1.b, XN8, SA, R, width, width, width, 1e-5f, s));
        FERN_TRY(run_gemm_b(c, gemm_desc_f8(XN8, SA, width, kv, QKV + width, 3 * width, (int)R, EPI_BIAS, false), s));
        HIP_TRY(launch_gather_rows(X, width, T1, width, batch, width, 1, S, 0, nullptr, s));
        HIP_TRY(launch_layernorm(T1, nullptr, Bk.ln1.g, Bk.ln1.b, T0, batch, width, width, width, 1e-5f, s));
    } else if (c->precision == FERN_PREC_BF16 || c->precision == FERN_PREC_MX8_MLP || c->precision == FERN_PREC_MX8_IMG) {
        // perf mode: the token-level K/V projection takes bf16 operands; the class-row chain below stays fp32
        unsigned short* XNb = reinterpret_cast<unsigned short*>(XN);
        HIP_TRY(launch_layernorm_bf16(X, Bk.ln1.g, Bk.ln1.b, XNb, R, width, width, width, 1e-5f, s));
        FERN_TRY(run_gemm_b(c, gemm_desc_b(XNb, width, kv, QKV + width, 3 * width, (int)R, EPI_BIAS, false), s));
        HIP_TRY(launch_gather_rows(X, width, T1, width, batch, width, 1, S, 0, nullptr, s));           // x[:, 0]
        HIP_TRY(launch_layernorm(T1, nullptr, Bk.ln1.g, Bk.ln1.b, T0, batch, width, width, width, 1e-5f, s));   // ln_1(x)[:, 0]
    } else {
        HIP_TRY(launch_layernorm(X, nullptr, Bk.ln1.g, Bk.ln1.b, XN, R, width, width, width, 1e-5f, s));
        FERN_TRY(run_gemm(c, gemm_desc(XN, width, kv, QKV + width, 3 * width, (int)R, EPI_BIAS), s));  // K, V for all tokens
        HIP_TRY(launch_gather_rows(XN, width, T0, width, batch, width, 1, S, 0, nullptr, s));          // ln_1(x)[:, 0]
    }
    LinearW qw{Bk.qkv.w, Bk.qkv.b, width, width};
    FERN_TRY(run_gemm(c, gemm_desc(T0, width, qw, T1, width, batch, EPI_BIAS), s));                    // Q for the class rows
    // one query per (batch, head): q rows are [batch, 1]; K/V are read in place from the packed buffer
    AttnParams a{T1, QKV + width, QKV + 2 * width, T0, (long)width, 3L * width, 3L * width, (long)width,
                 batch, heads, hd, 1, S, 0, 1.0f / std::sqrt((float)hd)};
    FERN_TRY(run_attention(c, a, s));
    if (Xb) HIP_TRY(launch_gather_rows_bf16(Xb, width, CLS, width, batch, width, S, s));               // residual x[:, 0]
    else HIP_TRY(launch_gather_rows(X, width, CLS, width, batch, width, 1, S, 0, nullptr, s));
    GemmParams po = gemm_desc(T0, width, Bk.out, CLS, width, batch, EPI_BIAS_RESIDUAL);
    po.R = CLS;
    FERN_TRY(run_gemm(c, po, s));
    HIP_TRY(launch_layernorm(CLS, nullptr, Bk.ln2.g, Bk.ln2.b, T0, batch, width, width, width, 1e-5f, s));
    FERN_TRY(run_gemm(c, gemm_desc(T0, width, Bk.fc, H, Bk.fc.out, batch, EPI_BIAS_GELU), s));
    GemmParams p2 = gemm_desc(H, Bk.fc.out, Bk.proj, CLS, width, batch, EPI_BIAS_RESIDUAL);
    p2.R = CLS;
    // The class rows' c_proj is M = batch with an unsplit k chain of 3072 (ViT-B): 24 workgroups walking 48 k tiles each, 58 us.  Like
    // the query-side combiners' hidden layer (CombinerW.ksplit) K is cut into 512-wide slices at THIS call site, for every batch size
    // (kernels.h: GemmParams.ksplit): 144 workgroups, ~12 us + a 5 us reduce that adds the slices in ascending order.
    const int ks = (Bk.fc.out >= 2048 && Bk.fc.out % 512 == 0 && (width & 3) == 0) ? Bk.fc.out / 512 : 1;
    if (ks > 1) {
        float* kpart;
        FERN_TRY(ws_get(c, (size_t)ks * batch * width, &kpart));
        p2.ksplit = ks; p2.kpart = kpart; p2.epi = EPI_BIAS; p2.R = nullptr;      // slices store raw sums; bias + residual happen in the reduce
        FERN_TRY(run_gemm(c, p2, s));
        HIP_TRY(launch_splitk_bias_residual(kpart, ks, batch, width, Bk.proj.b, CLS, width, CLS, width, s));
        return FERN_OK;
    }
    return run_gemm(c, p2, s);
}

// Front of the ViT tower: conv1 over the patches (its epilogue adds the positional embedding and skips the class slot) and the class token.
// H is free until the first block: the reduced-precision modes stage their rounded patch rows there.
static int vit_front(fern_ctx* c, const float* images, float* X, float* H, int b, hipStream_t s) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int vw = cf.v_width, g = cf.image_size / cf.patch_size, g2 = g * g, S = g2 + 1;
    if (c->precision == FERN_PREC_MX8 && W.conv_mx.wm && (3 * cf.patch_size * cf.patch_size) <= 1280) {
        // block-scaled mode: patch rows are quantised once (H is free until the first block) and conv1 runs on the scaled MFMA;
        // same epilogue (positional embedding added, class slot skipped)
        const int kd = 3 * cf.patch_size * cf.patch_size;
        const long rows = (long)b * g2;
        unsigned char* A8 = reinterpret_cast<unsigned char*>(H);
        unsigned char* SA = A8 + ((size_t)rows * kd + 255) / 256 * 256;
        HIP_TRY(launch_im2col_mx8(images, A8, SA, rows, b, cf.image_size, cf.patch_size, g, s));
        GemmParams pm = gemm_desc_mx(A8, SA, rows, kd, W.conv_mx, X, vw, (int)rows, EPI_PATCH_EMBED, false);
        pm.aux0 = W.vpos; pm.grid = g;
        FERN_TRY(run_gemm_b(c, pm, s));
    } else if ((c->precision == FERN_PREC_BF16 || c->precision == FERN_PREC_MX8_MLP || c->precision == FERN_PREC_MX8_IMG) && W.conv_mx.wb &&
               (3 * cf.patch_size * cf.patch_size) <= 1280) {
        // bf16-operand modes (round 6): patch rows rounded to bf16 once (H is free until the first block), conv1 on the bf16 MFMA; same epilogue
        const int kd = 3 * cf.patch_size * cf.patch_size;
        const long rows = (long)b * g2;
        unsigned short* Ab = reinterpret_cast<unsigned short*>(H);
        HIP_TRY(launch_im2col_bf16(images, Ab, b, cf.image_size, cf.patch_size, g, s));
        GemmParams pb = gemm_desc_b(Ab, kd, W.conv_mx, X, vw, (int)rows, EPI_PATCH_EMBED, false);
        pb.aux0 = W.vpos; pb.grid = g;
        FERN_TRY(run_gemm_b(c, pb, s));
    } else {
    // conv1 as an im2col-free GEMM; epilogue adds the positional embedding and skips the class slot
    GemmParams pe{};
    pe.A = images; pe.W = W.conv_w; pe.ldw = 3L * cf.patch_size * cf.patch_size; pe.C = X; pe.ldc = vw;
    pe.M = b * g2; pe.N = vw; pe.K = 3 * cf.patch_size * cf.patch_size;
    pe.epi = EPI_PATCH_EMBED; pe.aload = ALOAD_IM2COL; pe.aux0 = W.vpos;
    pe.img = cf.image_size; pe.patch = cf.patch_size; pe.grid = g;
    FERN_TRY(run_gemm(c, pe, s));
    }
    HIP_TRY(launch_vit_cls(W.cls, W.vpos, X, b, S, vw, s));
    return FERN_OK;
}

// Tail of the ViT tower: the last block for the class rows only, ln_post, the visual projection (modeling_clip.py:876-887).  ATT / H are
// free after the last full block: their heads are the [b, width] / [b, mlp] temporaries.  Xb: the bf16 residual stream of FERN_PREC_MX8.
static int vit_tail(fern_ctx* c, float* X, float* XN, float* QKV, float* ATT, float* H, float* CLS, float* out, int b, hipStream_t s,
                    const unsigned short* Xb = nullptr) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int vw = cf.v_width, g = cf.image_size / cf.patch_size, S = g * g + 1;
    FERN_TRY(clip_block_cls_only(c, W.vblocks[cf.v_layers - 1], X, XN, QKV, CLS, ATT, ATT + (size_t)b * vw, H, b, S, vw, cf.v_heads, s, Xb));
    HIP_TRY(launch_layernorm(CLS, nullptr, W.ln_post.g, W.ln_post.b, CLS, b, vw, vw, vw, 1e-5f, s));
    LinearW proj{W.vproj_t, nullptr, cf.embed_dim, vw};
    return run_gemm(c, gemm_desc(CLS, vw, proj, out, cf.embed_dim, b, EPI_BIAS), s);
}
// Tail of the text tower: ln_final over every token, the text projection of all rows (out_seq; global == seq[EOT]) or of the EOT rows only
// (modeling_clip.py:760-768, models/clip_model.py:23-31).
static int text_tail(fern_ctx* c, const float* X, float* XN, const int* eot, float* out_global, float* out_seq, int B, hipStream_t s) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int tw = cf.t_width, T = cf.context_length, E = cf.embed_dim;
    const long R = (long)B * T;
    HIP_TRY(launch_layernorm(X, nullptr, W.ln_final.g, W.ln_final.b, XN, R, tw, tw, tw, 1e-5f, s));
    LinearW proj{W.tproj_t, nullptr, E, tw};
    if (out_seq) {
        FERN_TRY(run_gemm(c, gemm_desc(XN, tw, proj, out_seq, E, (int)R, EPI_BIAS), s));
        if (out_global) HIP_TRY(launch_gather_rows(out_seq, E, out_global, E, B, E, 1, T, 0, eot, s));   // global == seq[EOT]
    } else if (out_global) {
        float* pooled;
        FERN_TRY(ws_get(c, (size_t)B * tw, &pooled));
        HIP_TRY(launch_gather_rows(XN, tw, pooled, tw, B, tw, 1, T, 0, eot, s));
        FERN_TRY(run_gemm(c, gemm_desc(pooled, tw, proj, out_global, E, B, EPI_BIAS), s));
    }
    return FERN_OK;
}

static int vit_chunk(fern_ctx* c, const float* images, float* out, int b, hipStream_t s) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int vw = cf.v_width, g = cf.image_size / cf.patch_size, g2 = g * g, S = g2 + 1;
    const long R = (long)b * S;
    float *X, *XN, *QKV, *ATT, *H, *CLS;
    FERN_TRY(ws_get(c, (size_t)R * vw, &X));
    FERN_TRY(ws_get(c, (size_t)R * vw, &XN));
    FERN_TRY(ws_get(c, (size_t)R * 3 * vw, &QKV));
    FERN_TRY(ws_get(c, (size_t)R * vw, &ATT));
    FERN_TRY(ws_get(c, (size_t)R * cf.v_mlp, &H));
    FERN_TRY(ws_get(c, (size_t)b * vw, &CLS));
    FERN_TRY(vit_front(c, images, X, H, b, s));
    unsigned short* Xb = nullptr;              // FERN_PREC_MX8: the bf16 residual stream of the full blocks
    const bool mx_stream = c->precision == FERN_PREC_MX8 && cf.v_layers > 1;
    if (mx_stream) {
        FERN_TRY(ws_get(c, (size_t)R * vw, &Xb));
        HIP_TRY(launch_layernorm_bf16(X, W.ln_pre.g, W.ln_pre.b, Xb, R, vw, vw, vw, 1e-5f, s));      // ln_pre writes the stream (rounded once)
    } else {
        HIP_TRY(launch_layernorm(X, nullptr, W.ln_pre.g, W.ln_pre.b, X, R, vw, vw, vw, 1e-5f, s));
    }
    for (int l = 0; l + 1 < cf.v_layers; ++l) {
        if (c->precision == FERN_PREC_MX8) {    // same buffer plan as fp8 below; the scale area holds E8M0 bytes
            unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
            FERN_TRY(clip_block_mx8(c, W.vblocks[l], Xb, XN8, reinterpret_cast<unsigned char*>(XN + ((size_t)R * vw / 4 + 63) / 64 * 64),
                                    reinterpret_cast<unsigned short*>(QKV), reinterpret_cast<unsigned short*>(ATT),
                                    reinterpret_cast<unsigned char*>(ATT) + (size_t)R * vw * 2, reinterpret_cast<unsigned char*>(H),
                                    b, S, vw, cf.v_heads, 0, s));
        } else if (c->precision == FERN_PREC_FP8) {    // XN: fp8 rows + scales; ATT / H: bf16 output in the first half, its fp8 copy behind it
            unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
            FERN_TRY(clip_block_fp8(c, W.vblocks[l], X, XN8, XN + ((size_t)R * vw / 4 + 63) / 64 * 64, reinterpret_cast<unsigned short*>(QKV),
                                    reinterpret_cast<unsigned short*>(ATT), reinterpret_cast<unsigned char*>(ATT) + (size_t)R * vw * 2,
                                    reinterpret_cast<unsigned short*>(H), reinterpret_cast<unsigned char*>(H) + (size_t)R * cf.v_mlp * 2,
                                    b, S, vw, cf.v_heads, 0, s));
        } else if (c->precision == FERN_PREC_MX8_MLP || c->precision == FERN_PREC_MX8_IMG) {
            FERN_TRY(clip_block_mxmlp(c, W.vblocks[l], X, XN, reinterpret_cast<unsigned short*>(QKV), reinterpret_cast<unsigned short*>(ATT),
                                      reinterpret_cast<unsigned char*>(H), b, S, vw, cf.v_heads, 0, s, c->precision == FERN_PREC_MX8_IMG));
        } else if (c->precision == FERN_PREC_BF16)     // XN / ATT / H double as the bf16 operand buffers (half filled)
            FERN_TRY(clip_block_bf16(c, W.vblocks[l], X, reinterpret_cast<unsigned short*>(XN), reinterpret_cast<unsigned short*>(QKV),
                                     reinterpret_cast<unsigned short*>(ATT),
                                     reinterpret_cast<unsigned short*>(H), b, S, vw, cf.v_heads, 0, s));
        else
            FERN_TRY(clip_block(c, W.vblocks[l], X, XN, QKV, ATT, H, b, S, vw, cf.v_heads, 0, s));
    }
    // the class-row chain of the last block is fp32; with the bf16 stream only the class rows are widened (exactly), inside the block
    return vit_tail(c, X, XN, QKV, ATT, H, CLS, out, b, s, mx_stream ? Xb : nullptr);
}

// ---- open_clip ModifiedResNet (RN50x4).  Activations are NHWC, so every 1x1 convolution is a plain GEMM over the pixel
// rows, every 3x3 convolution the same GEMM with the 3x3-window A loader, BatchNorm is folded into the weights and the
// ReLU / residual add live in the GEMM epilogue.
static GemmParams conv_desc(const float* A, const ConvW& W, float* C, int M, int epi) {
    GemmParams p{};
    p.A = A; p.lda = W.k; p.W = W.w; p.ldw = W.k; p.bias = W.b; p.C = C; p.ldc = W.cout;
    p.M = M; p.N = W.cout; p.K = W.k; p.epi = epi; p.aload = ALOAD_PLAIN;
    return p;
}
static int conv3x3(fern_ctx* c, const float* A, const ConvW& W, float* C, int b, int H, int Wd, int cin, hipStream_t s) {
    GemmParams p = conv_desc(A, W, C, b * H * Wd, EPI_BIAS_RELU);
    p.aload = ALOAD_CONV3; p.conv_h = H; p.conv_w = Wd; p.conv_c = cin; p.zeros = c->clip.res.zeros;
    return run_gemm(c, p, s);
}

static int resnet_chunk(fern_ctx* c, const float* images, float* out, int b, hipStream_t s) {
    const ClipW& W = c->clip;
    const ResNetW& R = W.res;
    const fern_clip_config& cf = W.cfg;
    const int S = cf.image_size, s1 = S / 2, w = cf.r_width;
    // one arena slot sized for the largest activation of the tower, six of them reused round-robin
    size_t max_act = (size_t)b * s1 * s1 * std::max(R.stem_c, w);
    {
        int H = S / 4;
        for (const auto& B : R.blocks) {
            max_act = std::max(max_act, (size_t)b * H * H * std::max(B.cin, B.planes * 4));
            if (B.stride > 1) H /= 2;
        }
    }
    {
        const int hf = S / 32;
        max_act = std::max(max_act, (size_t)b * (hf * hf + 1) * 2 * (size_t)(w * 32));      // attention-pool K/V
    }
    float* buf[6];
    for (auto& p : buf) FERN_TRY(ws_get(c, max_act, &p));
    // stem
    HIP_TRY(launch_stem_conv(images, R.stem1.w, R.stem1.b, buf[0], b, S, R.stem_c, s));
    FERN_TRY(conv3x3(c, buf[0], R.stem2, buf[1], b, s1, s1, R.stem_c, s));
    FERN_TRY(conv3x3(c, buf[1], R.stem3, buf[0], b, s1, s1, R.stem_c, s));
    HIP_TRY(launch_avgpool_nhwc(buf[0], buf[1], b, s1, s1, w, 2, s));
    float* X = buf[1];
    float* OUT = buf[0];
    int H = S / 4;
    for (const auto& B : R.blocks) {
        const int rows = b * H * H, Ho = H / B.stride, rows_o = b * Ho * Ho;
        float *T1 = buf[2], *T2 = buf[3], *TP = buf[4], *ID = buf[5];
        FERN_TRY(run_gemm(c, conv_desc(X, B.c1, T1, rows, EPI_BIAS_RELU), s));                    // 1x1 + BN + ReLU
        FERN_TRY(conv3x3(c, T1, B.c2, T2, b, H, H, B.planes, s));                                  // 3x3 + BN + ReLU
        const float* t2 = T2;
        const float* xin = X;
        if (B.stride > 1) {
            HIP_TRY(launch_avgpool_nhwc(T2, T1, b, H, H, B.planes, B.stride, s));                  // anti-aliasing avg-pool
            t2 = T1;
            HIP_TRY(launch_avgpool_nhwc(X, TP, b, H, H, B.cin, B.stride, s));
            xin = TP;
        }
        const float* identity = X;
        if (B.has_down) {
            FERN_TRY(run_gemm(c, conv_desc(xin, B.down, ID, rows_o, EPI_BIAS), s));                // shortcut: [avg-pool] + 1x1 + BN
            identity = ID;
        }
        GemmParams p3 = conv_desc(t2, B.c3, OUT, rows_o, EPI_BIAS_RESIDUAL_RELU);                  // 1x1 + BN + identity + ReLU
        p3.R = identity;
        FERN_TRY(run_gemm(c, p3, s));
        std::swap(X, OUT);
        H = Ho;
    }
    // AttentionPool2d: NHWC rows of one image are already its HW tokens
    const int HW = H * H, E = w * 32, heads = cf.r_heads, hd = E / heads;
    float *T = buf[2], *T0 = buf[3], *MEAN = buf[3] + (size_t)b * E, *KV = buf[4], *Q = buf[5], *ATT = buf[5] + (size_t)b * E;
    HIP_TRY(launch_attnpool_tokens(X, MEAN, R.pos, T, T0, b, HW, E, s));
    FERN_TRY(run_gemm(c, gemm_desc(T, E, R.kv, KV, 2 * E, b * (HW + 1), EPI_BIAS), s));
    FERN_TRY(run_gemm(c, gemm_desc(T0, E, R.q, Q, E, b, EPI_BIAS), s));
    AttnParams a{Q, KV, KV + E, ATT, (long)E, 2L * E, 2L * E, (long)E, b, heads, hd, 1, HW + 1, 0, 1.0f / std::sqrt((float)hd)};
    FERN_TRY(run_attention(c, a, s));
    return run_gemm(c, gemm_desc(ATT, E, R.cproj, out, cf.embed_dim, b, EPI_BIAS), s);
}

extern "C" int fern_vit_encode_image(fern_ctx* c, const float* images, float* out, int b, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_vit_encode_image: ctx is NULL");
    if (!c->clip.ready || (c->clip.cfg.v_arch == 0 && c->clip.cfg.v_layers <= 0))
        return fail(FERN_ERR_STATE, "fern_vit_encode_image: image tower not finalised (fern_finalize_clip)");
    FERN_TRY(check_fresh(c, "fern_vit_encode_image"));
    if (b < 0 || (b && (!images || !out))) return fail(FERN_ERR_ARG, "fern_vit_encode_image: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const fern_clip_config& cf = c->clip.cfg;
    const long img_sz = 3L * cf.image_size * cf.image_size;
    const bool resnet = cf.v_arch == 1;
    const int CH = resnet ? 128 : 64;      // the ResNet's late stages have few pixels per image: larger chunks fill the chip (M = 128 x 81 rows)
    for (int o = 0; o < b; o += CH) {
        const int m = std::min(CH, b - o);
        FERN_TRY(ws_begin(c, s));
        if (resnet) FERN_TRY(resnet_chunk(c, images + o * img_sz, out + (long)o * cf.embed_dim, m, s));
        else FERN_TRY(vit_chunk(c, images + o * img_sz, out + (long)o * cf.embed_dim, m, s));
    }
    return FERN_OK;
}

static int text_chunk(fern_ctx* c, const int64_t* tokens, float* out_global, float* out_seq, int B, hipStream_t s) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int tw = cf.t_width, T = cf.context_length;
    const long R = (long)B * T;
    float *X, *XN, *QKV, *ATT, *H;
    int* eot;
    FERN_TRY(ws_get(c, (size_t)R * tw, &X));
    FERN_TRY(ws_get(c, (size_t)R * tw, &XN));
    FERN_TRY(ws_get(c, (size_t)R * 3 * tw, &QKV));
    FERN_TRY(ws_get(c, (size_t)R * tw, &ATT));
    FERN_TRY(ws_get(c, (size_t)R * cf.t_mlp, &H));
    FERN_TRY(ws_get(c, (size_t)B, &eot));
    HIP_TRY(launch_text_embed(tokens, W.tok_emb, W.tpos, X, eot, B, T, tw, cf.vocab_size, c->tok_flag, s));
    const bool text_mx = c->precision == FERN_PREC_MX8;
    unsigned short* Xb = nullptr;              // FERN_PREC_MX8: the bf16 residual stream of the blocks (clip_block_mx8)
    if (text_mx) {
        FERN_TRY(ws_get(c, (size_t)R * tw, &Xb));
        HIP_TRY(launch_f32_to_bf16(X, Xb, R * tw, s));
    }
    for (int l = 0; l < cf.t_layers; ++l) {
        if (text_mx) {
            unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
            FERN_TRY(clip_block_mx8(c, W.tblocks[l], Xb, XN8, reinterpret_cast<unsigned char*>(XN + ((size_t)R * tw / 4 + 63) / 64 * 64),
                                    reinterpret_cast<unsigned short*>(QKV), reinterpret_cast<unsigned short*>(ATT),
                                    reinterpret_cast<unsigned char*>(ATT) + (size_t)R * tw * 2, reinterpret_cast<unsigned char*>(H),
                                    B, T, tw, cf.t_heads, 1, s));
        } else if (c->precision == FERN_PREC_FP8) {
            unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
            FERN_TRY(clip_block_fp8(c, W.tblocks[l], X, XN8, XN + ((size_t)R * tw / 4 + 63) / 64 * 64, reinterpret_cast<unsigned short*>(QKV),
                                    reinterpret_cast<unsigned short*>(ATT), reinterpret_cast<unsigned char*>(ATT) + (size_t)R * tw * 2,
                                    reinterpret_cast<unsigned short*>(H), reinterpret_cast<unsigned char*>(H) + (size_t)R * cf.t_mlp * 2,
                                    B, T, tw, cf.t_heads, 1, s));
        } else if (c->precision == FERN_PREC_BF16 || c->precision == FERN_PREC_MX8_MLP || c->precision == FERN_PREC_MX8_IMG)
            // FERN_PREC_MX8_MLP / _IMG (round 6): the TEXT tower runs the bf16 block throughout -- its GEMMs (M = 77 B rows, K = 512) are
            // the ones the block-scaled kernels run worst (0.07-0.12 of their peak) and its rounding points sit directly on the query:
            // e4m3 there cost 0.9 pp of Recall@50 (bench.py `reduced_modes`: mx8mlp -1.03 -> -0.15 pp) for ~0.1 ms of a 4 ms step
            FERN_TRY(clip_block_bf16(c, W.tblocks[l], X, reinterpret_cast<unsigned short*>(XN), reinterpret_cast<unsigned short*>(QKV),
                                     reinterpret_cast<unsigned short*>(ATT),
                                     reinterpret_cast<unsigned short*>(H), B, T, tw, cf.t_heads, 1, s));
        else
            FERN_TRY(clip_block(c, W.tblocks[l], X, XN, QKV, ATT, H, B, T, tw, cf.t_heads, 1, s));
    }
    if (Xb) HIP_TRY(launch_bf16_to_f32(Xb, X, R * tw, s));
    return text_tail(c, X, XN, eot, out_global, out_seq, B, s);
}

// Both towers of a composed query in one pass (round 6), fp32 data flow only (FERN_PREC_FP32, also under f32x3): the
// image tower's full blocks and the text tower's blocks are walked LAYER BY LAYER and the four GEMMs of a layer -- QKV, out-proj, c_fc,
// c_proj -- are issued as image + text PAIRS (run_gemm_pair): the text layer's few hundred tiles ride in the image layer's launch and
// back-fill its tail instead of running as under-filled launches of their own.  Everything else (LayerNorms, attention, the class-row
// block, the projections) is issued exactly as vit_chunk / text_chunk issue it, and every GEMM computes the same tiles in the same k
// order: the features are bit-identical to fern_vit_encode_image + fern_text_encode.
static int pair_chunk(fern_ctx* c, const float* images, float* out_img, const int64_t* tokens, float* out_global, float* out_seq, int b, hipStream_t s) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int vw = cf.v_width, g = cf.image_size / cf.patch_size, g2 = g * g, S = g2 + 1;
    const int tw = cf.t_width, T = cf.context_length;
    const long R = (long)b * S, Rt = (long)b * T;
    float *X, *XN, *QKV, *ATT, *H, *CLS, *Xt, *XNt, *QKVt, *ATTt, *Ht;
    int* eot;
    FERN_TRY(ws_get(c, (size_t)R * vw, &X));
    FERN_TRY(ws_get(c, (size_t)R * vw, &XN));
    FERN_TRY(ws_get(c, (size_t)R * 3 * vw, &QKV));
    FERN_TRY(ws_get(c, (size_t)R * vw, &ATT));
    FERN_TRY(ws_get(c, (size_t)R * cf.v_mlp, &H));
    FERN_TRY(ws_get(c, (size_t)b * vw, &CLS));
    FERN_TRY(ws_get(c, (size_t)Rt * tw, &Xt));
    FERN_TRY(ws_get(c, (size_t)Rt * tw, &XNt));
    FERN_TRY(ws_get(c, (size_t)Rt * 3 * tw, &QKVt));
    FERN_TRY(ws_get(c, (size_t)Rt * tw, &ATTt));
    FERN_TRY(ws_get(c, (size_t)Rt * cf.t_mlp, &Ht));
    FERN_TRY(ws_get(c, (size_t)b, &eot));
    // image tower front: conv1 as an im2col-free GEMM (epilogue adds the positional embedding), class token, ln_pre -- as vit_chunk
    GemmParams pe{};
    pe.A = images; pe.W = W.conv_w; pe.ldw = 3L * cf.patch_size * cf.patch_size; pe.C = X; pe.ldc = vw;
    pe.M = b * g2; pe.N = vw; pe.K = 3 * cf.patch_size * cf.patch_size;
    pe.epi = EPI_PATCH_EMBED; pe.aload = ALOAD_IM2COL; pe.aux0 = W.vpos;
    pe.img = cf.image_size; pe.patch = cf.patch_size; pe.grid = g;
    FERN_TRY(run_gemm(c, pe, s));
    HIP_TRY(launch_vit_cls(W.cls, W.vpos, X, b, S, vw, s));
    HIP_TRY(launch_layernorm(X, nullptr, W.ln_pre.g, W.ln_pre.b, X, R, vw, vw, vw, 1e-5f, s));
    // text tower front -- as text_chunk
    HIP_TRY(launch_text_embed(tokens, W.tok_emb, W.tpos, Xt, eot, b, T, tw, cf.vocab_size, c->tok_flag, s));
    const int vfull = cf.v_layers - 1;                     // the image tower's last block runs for the class rows only
    const int paired = vfull < cf.t_layers ? vfull : cf.t_layers;
    const int vhd = vw / cf.v_heads, thd = tw / cf.t_heads;
    for (int l = 0; l < paired; ++l) {
        const ClipBlockW &Bv = W.vblocks[l], &Bt = W.tblocks[l];
        HIP_TRY(launch_layernorm(X, nullptr, Bv.ln1.g, Bv.ln1.b, XN, R, vw, vw, vw, 1e-5f, s));
        HIP_TRY(launch_layernorm(Xt, nullptr, Bt.ln1.g, Bt.ln1.b, XNt, Rt, tw, tw, tw, 1e-5f, s));
        FERN_TRY(run_gemm_pair(c, gemm_desc(XN, vw, Bv.qkv, QKV, 3 * vw, (int)R, EPI_BIAS), gemm_desc(XNt, tw, Bt.qkv, QKVt, 3 * tw, (int)Rt, EPI_BIAS), s));
        AttnParams av{QKV, QKV + vw, QKV + 2 * vw, ATT, 3L * vw, 3L * vw, 3L * vw, (long)vw, b, cf.v_heads, vhd, S, S, 0, 1.0f / std::sqrt((float)vhd)};
        FERN_TRY(run_attention(c, av, s));
        AttnParams at{QKVt, QKVt + tw, QKVt + 2 * tw, ATTt, 3L * tw, 3L * tw, 3L * tw, (long)tw, b, cf.t_heads, thd, T, T, 1, 1.0f / std::sqrt((float)thd)};
        FERN_TRY(run_attention(c, at, s));
        GemmParams pov = gemm_desc(ATT, vw, Bv.out, X, vw, (int)R, EPI_BIAS_RESIDUAL), pot = gemm_desc(ATTt, tw, Bt.out, Xt, tw, (int)Rt, EPI_BIAS_RESIDUAL);
        pov.R = X; pot.R = Xt;
        FERN_TRY(run_gemm_pair(c, pov, pot, s));
        HIP_TRY(launch_layernorm(X, nullptr, Bv.ln2.g, Bv.ln2.b, XN, R, vw, vw, vw, 1e-5f, s));
        HIP_TRY(launch_layernorm(Xt, nullptr, Bt.ln2.g, Bt.ln2.b, XNt, Rt, tw, tw, tw, 1e-5f, s));
        FERN_TRY(run_gemm_pair(c, gemm_desc(XN, vw, Bv.fc, H, Bv.fc.out, (int)R, EPI_BIAS_GELU), gemm_desc(XNt, tw, Bt.fc, Ht, Bt.fc.out, (int)Rt, EPI_BIAS_GELU), s));
        GemmParams ppv = gemm_desc(H, Bv.fc.out, Bv.proj, X, vw, (int)R, EPI_BIAS_RESIDUAL), ppt = gemm_desc(Ht, Bt.fc.out, Bt.proj, Xt, tw, (int)Rt, EPI_BIAS_RESIDUAL);
        ppv.R = X; ppt.R = Xt;
        FERN_TRY(run_gemm_pair(c, ppv, ppt, s));
    }
    for (int l = paired; l < vfull; ++l) FERN_TRY(clip_block(c, W.vblocks[l], X, XN, QKV, ATT, H, b, S, vw, cf.v_heads, 0, s));
    for (int l = paired; l < cf.t_layers; ++l) FERN_TRY(clip_block(c, W.tblocks[l], Xt, XNt, QKVt, ATTt, Ht, b, T, tw, cf.t_heads, 1, s));
    FERN_TRY(vit_tail(c, X, XN, QKV, ATT, H, CLS, out_img, b, s));
    return text_tail(c, Xt, XNt, eot, out_global, out_seq, b, s);
}

// The same walk for FERN_PREC_MX8_IMG (the c5 default): the image tower's block-scaled GEMM of a layer and the text tower's bf16 GEMM of the
// same kind as one launch (run_gemm_b_pair).  Everything else as vit_chunk (clip_block_mxmlp with attn_mx) / text_chunk (clip_block_bf16)
// issue it; bit-identical to the two entry points.
static int pair_chunk_mximg(fern_ctx* c, const float* images, float* out_img, const int64_t* tokens, float* out_global, float* out_seq, int b, hipStream_t s) {
    const ClipW& W = c->clip;
    const fern_clip_config& cf = W.cfg;
    const int vw = cf.v_width, g = cf.image_size / cf.patch_size, g2 = g * g, S = g2 + 1;
    const int tw = cf.t_width, T = cf.context_length;
    const long R = (long)b * S, Rt = (long)b * T;
    float *X, *XN, *QKV, *ATT, *H, *CLS, *Xt, *XNt, *QKVt, *ATTt, *Ht;
    int* eot;
    FERN_TRY(ws_get(c, (size_t)R * vw, &X));
    FERN_TRY(ws_get(c, (size_t)R * vw, &XN));
    FERN_TRY(ws_get(c, (size_t)R * 3 * vw, &QKV));
    FERN_TRY(ws_get(c, (size_t)R * vw, &ATT));
    FERN_TRY(ws_get(c, (size_t)R * cf.v_mlp, &H));
    FERN_TRY(ws_get(c, (size_t)b * vw, &CLS));
    FERN_TRY(ws_get(c, (size_t)Rt * tw, &Xt));
    FERN_TRY(ws_get(c, (size_t)Rt * tw, &XNt));
    FERN_TRY(ws_get(c, (size_t)Rt * 3 * tw, &QKVt));
    FERN_TRY(ws_get(c, (size_t)Rt * tw, &ATTt));
    FERN_TRY(ws_get(c, (size_t)Rt * cf.t_mlp, &Ht));
    FERN_TRY(ws_get(c, (size_t)b, &eot));
    FERN_TRY(vit_front(c, images, X, H, b, s));
    HIP_TRY(launch_layernorm(X, nullptr, W.ln_pre.g, W.ln_pre.b, X, R, vw, vw, vw, 1e-5f, s));
    HIP_TRY(launch_text_embed(tokens, W.tok_emb, W.tpos, Xt, eot, b, T, tw, cf.vocab_size, c->tok_flag, s));
    const int vfull = cf.v_layers - 1;
    const int paired = vfull < cf.t_layers ? vfull : cf.t_layers;
    const int vhd = vw / cf.v_heads, thd = tw / cf.t_heads, mlp = cf.v_mlp;
    // operand views of the image tower's buffers (clip_block_mxmlp) and the text tower's (clip_block_bf16)
    unsigned char* XN8 = reinterpret_cast<unsigned char*>(XN);
    unsigned char* SM = reinterpret_cast<unsigned char*>(XN + ((size_t)R * vw / 4 + 63) / 64 * 64);
    unsigned char* SH = SM + ((size_t)R * (vw / 32) + 255) / 256 * 256;
    unsigned short* QKVb = reinterpret_cast<unsigned short*>(QKV);
    unsigned short* ATTb = reinterpret_cast<unsigned short*>(ATT);
    unsigned char* ATT8 = reinterpret_cast<unsigned char*>(ATT) + (size_t)R * vw * 2;
    unsigned char* H8 = reinterpret_cast<unsigned char*>(H);
    unsigned short *XNtb = reinterpret_cast<unsigned short*>(XNt), *QKVtb = reinterpret_cast<unsigned short*>(QKVt),
                   *ATTtb = reinterpret_cast<unsigned short*>(ATTt), *Htb = reinterpret_cast<unsigned short*>(Ht);
    for (int l = 0; l < paired; ++l) {
        const ClipBlockW &Bv = W.vblocks[l], &Bt = W.tblocks[l];
        HIP_TRY(launch_layernorm_mx8(X, Bv.ln1.g, Bv.ln1.b, XN8, SM, R, R, vw, vw, vw, 1e-5f, s));
        HIP_TRY(launch_layernorm_bf16(Xt, Bt.ln1.g, Bt.ln1.b, XNtb, Rt, tw, tw, tw, 1e-5f, s));
        FERN_TRY(run_gemm_b_pair(c, gemm_desc_mx(XN8, SM, R, vw, Bv.qkv, QKVb, 3 * vw, (int)R, EPI_BIAS, true),
                                 gemm_desc_b(XNtb, tw, Bt.qkv, QKVtb, 3 * tw, (int)Rt, EPI_BIAS, true), s));
        AttnParams av{nullptr, nullptr, nullptr, nullptr, 3L * vw, 3L * vw, 3L * vw, (long)vw,
                      b, cf.v_heads, vhd, S, S, 0, 1.0f / std::sqrt((float)vhd), ATTb, QKVb, QKVb + vw, QKVb + 2 * vw};
        av.out_b = nullptr; av.out_q8 = ATT8; av.out_scales = SM; av.out_srows = R;
        FERN_TRY(run_attention(c, av, s));
        AttnParams at{nullptr, nullptr, nullptr, nullptr, 3L * tw, 3L * tw, 3L * tw, (long)tw,
                      b, cf.t_heads, thd, T, T, 1, 1.0f / std::sqrt((float)thd), ATTtb, QKVtb, QKVtb + tw, QKVtb + 2 * tw};
        FERN_TRY(run_attention(c, at, s));
        GemmParams pov = gemm_desc_mx(ATT8, SM, R, vw, Bv.out, X, vw, (int)R, EPI_BIAS_RESIDUAL, false);
        GemmParams pot = gemm_desc_b(ATTtb, tw, Bt.out, Xt, tw, (int)Rt, EPI_BIAS_RESIDUAL, false);
        pov.R = X; pot.R = Xt;
        FERN_TRY(run_gemm_b_pair(c, pov, pot, s));
        HIP_TRY(launch_layernorm_mx8(X, Bv.ln2.g, Bv.ln2.b, XN8, SM, R, R, vw, vw, vw, 1e-5f, s));
        HIP_TRY(launch_layernorm_bf16(Xt, Bt.ln2.g, Bt.ln2.b, XNtb, Rt, tw, tw, tw, 1e-5f, s));
        GemmParams pfv = gemm_desc_mx(XN8, SM, R, vw, Bv.fc, H8, mlp, (int)R, EPI_BIAS_GELU, false);
        pfv.out_mx8 = 1; pfv.mxc = SH; pfv.mxc_rows = R;
        FERN_TRY(run_gemm_b_pair(c, pfv, gemm_desc_b(XNtb, tw, Bt.fc, Htb, Bt.fc.out, (int)Rt, EPI_BIAS_GELU, true), s));
        GemmParams ppv = gemm_desc_mx(H8, SH, R, mlp, Bv.proj, X, vw, (int)R, EPI_BIAS_RESIDUAL, false);
        GemmParams ppt = gemm_desc_b(Htb, Bt.fc.out, Bt.proj, Xt, tw, (int)Rt, EPI_BIAS_RESIDUAL, false);
        ppv.R = X; ppt.R = Xt;
        FERN_TRY(run_gemm_b_pair(c, ppv, ppt, s));
    }
    for (int l = paired; l < vfull; ++l)
        FERN_TRY(clip_block_mxmlp(c, W.vblocks[l], X, XN, QKVb, ATTb, H8, b, S, vw, cf.v_heads, 0, s, true));
    for (int l = paired; l < cf.t_layers; ++l) FERN_TRY(clip_block_bf16(c, W.tblocks[l], Xt, XNtb, QKVtb, ATTtb, Htb, b, T, tw, cf.t_heads, 1, s));
    FERN_TRY(vit_tail(c, X, XN, QKV, ATT, H, CLS, out_img, b, s));
    return text_tail(c, Xt, XNt, eot, out_global, out_seq, b, s);
}

// A token id outside the vocabulary cannot raise from inside a kernel (nn.Embedding does, in the reference): the embedding
// kernel poisons the row with NaN and sets a host-mapped flag; it is turned into FERN_ERR_ARG here, at fern_sync and at the
// next fern_text_encode -- without a synchronisation on the launch path.
static int check_token_flag(fern_ctx* c, const char* fn) {
    if (!c->tok_flag) return FERN_OK;
    const int v = __atomic_load_n(c->tok_flag, __ATOMIC_RELAXED);
    if (v == 0) return FERN_OK;
    __atomic_store_n(c->tok_flag, 0, __ATOMIC_RELAXED);
    return fail(FERN_ERR_ARG, std::string(fn) + ": an earlier fern_text_encode on this context read a token id outside [0, vocab_size) at flat "
                                                 "position " + std::to_string(v - 1) + " of its chunk (its features are NaN): tokenizer / vocabulary mismatch?");
}

extern "C" int fern_text_encode(fern_ctx* c, const int64_t* tokens, const float* visual_emb, const int64_t* visual_emb_shape, float* out_global,
                                float* out_seq, int B, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_text_encode: ctx is NULL");
    if (!c->clip.ready || c->clip.cfg.t_layers <= 0) return fail(FERN_ERR_STATE, "fern_text_encode: text tower not finalised (fern_finalize_clip)");
    FERN_TRY(check_fresh(c, "fern_text_encode"));
    if (B < 0 || (B && (!tokens || (!out_global && !out_seq)))) return fail(FERN_ERR_ARG, "fern_text_encode: bad argument");
    const fern_clip_config& cf = c->clip.cfg;
    if (visual_emb) {      // clip_model.py:23-31: visual_emb = ref_patch_feats.transpose(0, 1), [13, B, D]; checked, then unused (vanilla CLIP branch)
        if (!visual_emb_shape || visual_emb_shape[0] != 13 || visual_emb_shape[1] != B || visual_emb_shape[2] != cf.embed_dim)
            return fail(FERN_ERR_ARG, "fern_text_encode: visual_emb must be [13, B, embed_dim]");
    }
    HIP_TRY(hipSetDevice(c->device));
    if (!c->tok_flag) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->tok_flag), sizeof(int), hipHostMallocMapped));
        *c->tok_flag = 0;
    }
    FERN_TRY(check_token_flag(c, "fern_text_encode"));
    hipStream_t s = (hipStream_t)stream;
    const int CH = 256;
    for (int o = 0; o < B; o += CH) {
        const int m = std::min(CH, B - o);
        FERN_TRY(ws_begin(c, s));
        FERN_TRY(text_chunk(c, tokens + (long)o * cf.context_length, out_global ? out_global + (long)o * cf.embed_dim : nullptr,
                            out_seq ? out_seq + (long)o * cf.context_length * cf.embed_dim : nullptr, m, s));
    }
    return FERN_OK;
}

extern "C" int fern_encode_pair(fern_ctx* c, const float* images, const int64_t* tokens, float* out_image, float* out_global, float* out_seq, int B,
                                void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_encode_pair: ctx is NULL");
    if (!c->clip.ready) return fail(FERN_ERR_STATE, "fern_encode_pair: CLIP weights not finalised (fern_finalize_clip)");
    FERN_TRY(check_fresh(c, "fern_encode_pair"));
    if (B < 0 || (B && (!images || !tokens || !out_image || (!out_global && !out_seq)))) return fail(FERN_ERR_ARG, "fern_encode_pair: bad argument");
    const fern_clip_config& cf = c->clip.cfg;
    const bool towers = cf.v_arch == 0 && cf.v_layers > 1 && cf.t_layers > 0;
    // the mixed mode pairs when every image GEMM of a block is block-scaled (head dim a multiple of 32: clip_block_mxmlp) and the text GEMMs
    // meet the bf16 family's k tiling
    const bool pair_mx = towers && c->precision == FERN_PREC_MX8_IMG && (cf.v_width / cf.v_heads) % 32 == 0 && cf.v_width % 128 == 0 &&
                         cf.v_mlp % 128 == 0 && cf.t_width % 32 == 0 && cf.t_mlp % 32 == 0;
    const bool pairable = towers && (c->precision == FERN_PREC_FP32 || pair_mx);
    if (!pairable) {      // every other mode / tower: the two entry points, one after the other (same results by definition)
        FERN_TRY(fern_vit_encode_image(c, images, out_image, B, stream));
        return fern_text_encode(c, tokens, nullptr, nullptr, out_global, out_seq, B, stream);
    }
    HIP_TRY(hipSetDevice(c->device));
    if (!c->tok_flag) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->tok_flag), sizeof(int), hipHostMallocMapped));
        *c->tok_flag = 0;
    }
    FERN_TRY(check_token_flag(c, "fern_encode_pair"));
    hipStream_t s = (hipStream_t)stream;
    const long img_sz = 3L * cf.image_size * cf.image_size;
    const int CH = 64;                                      // vit_chunk's chunk: the pairing keeps the image tower's launch shapes
    for (int o = 0; o < B; o += CH) {
        const int m = std::min(CH, B - o);
        FERN_TRY(ws_begin(c, s));
        FERN_TRY((pair_mx ? pair_chunk_mximg : pair_chunk)(c, images + o * img_sz, out_image + (long)o * cf.embed_dim, tokens + (long)o * cf.context_length,
                            out_global ? out_global + (long)o * cf.embed_dim : nullptr,
                            out_seq ? out_seq + (long)o * cf.context_length * cf.embed_dim : nullptr, m, s));
    }
    return FERN_OK;
}

// ------------------------------------------------------------------------------------------------
// rank
// ------------------------------------------------------------------------------------------------
// Fused sweep + selection (kernels.h: TopkFilter): plan and workspace of one query chunk.
//   sample pass   S = max(N / 64, min(N, 4096)) rows, one per run of R = N / S rows, scores stored [m, S]   -> PROF_TOPK
//   bound         per query the K-th best sample key, a lower bound of the true K-th best                   -> PROF_TOPK
//   sweep         the full pass: nothing stored, ~K*R survivors per query appended to its 256 lists         -> PROF_SWEEP
//   select        exact top-K of the lists; a query with an overflowed list is handed to the exact pass     -> PROF_TOPK
//   exact pass    gated on flags[0] (an empty launch unless a list overflowed): streams the gallery for the
//                 flagged queries through sorted wave lists -- no capacity, so the stage is exact whatever
//                 the gallery looks like                                                                    -> PROF_TOPK
struct RankPlan {
    long S; int R; int cap; int groups;
    float* sample; long ld;
    unsigned long long* thr; int* count; int* flags; int* state; int* done;
    unsigned long long* partial;
    TopkFilter filt;
};
static int rank_plan(fern_ctx* c, int m, int64_t N, int K, const int32_t* exclude, int64_t idx_offset, RankPlan* P, long sample_rows = 0) {
    P->S = sample_rows > 0 ? sample_rows : std::min<long>(std::max<long>(N / 64, std::min<long>(N, 4096)), 32768);      // the bound kernel holds a query's sample in registers
    P->R = P->S > 0 ? (int)(N / P->S) : 1;
    // a list sees ~K * R / 256 survivors (<= 16 at K = R = 64; beyond N = 2M rows R grows past 64 and lists start to overflow:
    // those queries are then ranked by the exact pass -- still exact, a gallery pass slower); 64 entries is what one wave load of
    // the select kernel covers
    P->cap = 64;
    static const int cap_override = [] { const char* e = std::getenv("FERN_RANK_CAP"); return e ? std::atoi(e) : 0; }();
    if (cap_override >= 1 && cap_override <= 64) P->cap = cap_override;     // test hook: tiny lists force the overflow -> exact pass path
    P->ld = (std::max<long>(P->S, 4) + 3) & ~3L;
    P->groups = (int)std::min<long>(256, std::max<long>(1, (N + 4095) / 4096));
    FERN_TRY(ws_get(c, (size_t)m * P->ld, &P->sample));
    FERN_TRY(ws_get(c, (size_t)m, &P->thr));
    FERN_TRY(ws_get(c, (size_t)m * RANK_SLOTS, &P->count));
    FERN_TRY(ws_get(c, (size_t)4, &P->flags));
    FERN_TRY(ws_get(c, (size_t)2 * m, &P->state));
    P->done = P->state + m;
    FERN_TRY(ws_get(c, (size_t)m * P->groups * 64, &P->partial));
    unsigned long long* cand;
    FERN_TRY(ws_get(c, (size_t)m * RANK_SLOTS * P->cap, &cand));
    P->filt = TopkFilter{cand, P->thr, P->count, exclude, (long)idx_offset, P->cap};
    return FERN_OK;
}
static const size_t kRankQueryChunk = 1024;      // queries per plan: bounds cand[m][256][64] (128 KiB per query)

extern "C" int fern_sim_topk(fern_ctx* c, const float* q, const float* gallery, int B, int64_t N, int D, int K, float* out_scores,
                             int32_t* out_idx, int64_t idx_offset, const int32_t* exclude_idx, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_sim_topk: ctx is NULL");
    if (B < 0 || N < 0 || K < 1 || K > 64 || D <= 0 || D % 32) return fail(FERN_ERR_ARG, "fern_sim_topk: need 1<=K<=64, D % 32 == 0");
    if (B && (!q || !out_scores || !out_idx || (N && !gallery))) return fail(FERN_ERR_ARG, "fern_sim_topk: NULL argument");
    if (N > 0x7FFFFFF0LL) return fail(FERN_ERR_ARG, "fern_sim_topk: N too large for int32 indices");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return FERN_OK;
    for (long o = 0; o < B; o += (long)kRankQueryChunk) {
        const int m = (int)std::min<long>((long)kRankQueryChunk, B - o);
        const int32_t* ex = exclude_idx ? exclude_idx + o : nullptr;
        FERN_TRY(ws_begin(c, s));
        RankPlan P;
        FERN_TRY(rank_plan(c, m, N, K, ex, idx_offset, &P));
        int slot, stage;
        FERN_TRY(prof_open(c, PROF_STAGE, 0, s, &stage));      // the whole stage of this chunk: one marker pair (round 4: three pairs charged their overhead to a ~90 us stage)
        if (P.S > 0) {      // sample pass: the same GEMM, W rows = jittered 1-in-R sample of the gallery
            GemmParams p{};
            p.A = q + o * D; p.lda = D; p.W = gallery; p.ldw = D; p.C = P.sample; p.ldc = P.ld;
            p.M = m; p.N = (int)P.S; p.K = D; p.epi = EPI_BIAS; p.aload = ALOAD_PLAIN; p.w_sample = P.R;
            HIP_TRY(launch_gemm(p, s));
        }
        HIP_TRY(launch_topk_sample_bound(P.sample, P.ld, m, P.S, P.R, K, ex, idx_offset, P.thr, P.count, P.flags, P.state, s));
        GemmParams p{};
        p.A = q + o * D; p.lda = D; p.W = gallery; p.ldw = D; p.ldc = 4;
        p.M = m; p.N = (int)N; p.K = D; p.epi = EPI_TOPK_FILTER; p.aload = ALOAD_PLAIN; p.filt = P.filt;
        // algorithmic bytes of the sweep: gallery once, queries, results
        if (N > 0) FERN_TRY(run_gemm(c, p, s, PROF_SWEEP, (double)N * D * 4 + (double)m * D * 4 + (double)m * K * 8));
        HIP_TRY(launch_topk_candidates(P.filt, m, K, idx_offset, out_scores + o * K, out_idx + o * K, P.flags, P.state, s));
        HIP_TRY(launch_rank_exact(q + o * D, gallery, 0, m, N, D, K, P.state, P.thr, ex, idx_offset, idx_offset, P.partial, P.groups, P.done,
                                  out_scores + o * K, out_idx + o * K, P.flags, s));
        FERN_TRY(prof_close(c, stage, s));
        (void)slot;
    }
    return FERN_OK;
}

// BASELINE config 5 ("bf16 similarity"): the gallery is stored in bf16 (half the HBM bytes), queries are rounded to bf16 in
// the kernel, products accumulate in fp32.  Same outputs and tie rule as fern_sim_topk.
extern "C" int fern_gallery_to_bf16(fern_ctx* c, const float* src, uint16_t* dst, int64_t n, int d, void* stream) {
    if (!c || n < 0 || d <= 0 || (n && (!src || !dst))) return fail(FERN_ERR_ARG, "fern_gallery_to_bf16: bad argument");
    if (d % 4) return fail(FERN_ERR_ARG, "fern_gallery_to_bf16: D must be a multiple of 4");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_f32_to_bf16(src, dst, (long)n * d, (hipStream_t)stream));
    return FERN_OK;
}

extern "C" int fern_sim_topk_bf16(fern_ctx* c, const float* q, const uint16_t* gallery, int B, int64_t N, int D, int K, float* out_scores,
                                  int32_t* out_idx, int64_t idx_offset, const int32_t* exclude_idx, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_sim_topk_bf16: ctx is NULL");
    if (B < 0 || N < 0 || K < 1 || K > 64 || D <= 0 || D % 64 || D > 1024) return fail(FERN_ERR_ARG, "fern_sim_topk_bf16: need 1<=K<=64, D % 64 == 0, D <= 1024");
    if (B && (!q || !out_scores || !out_idx || (N && !gallery))) return fail(FERN_ERR_ARG, "fern_sim_topk_bf16: NULL argument");
    if (N > 0x7FFFFFF0LL) return fail(FERN_ERR_ARG, "fern_sim_topk_bf16: N too large for int32 indices");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return FERN_OK;
    for (long o = 0; o < B; o += (long)kRankQueryChunk) {
        const int m = (int)std::min<long>((long)kRankQueryChunk, B - o);
        const int32_t* ex = exclude_idx ? exclude_idx + o : nullptr;
        FERN_TRY(ws_begin(c, s));
        RankPlan P;
        FERN_TRY(rank_plan(c, m, N, K, ex, idx_offset, &P));
        // one sweep launch per query block, shared plan buffers: 128 queries per gallery pass where the kernel has the two-block form
        // (D = 64 / 128 / 256 / 512: second block's bf16 image in LDS), else 64
        const long QBLK = (D == 64 || D == 128 || D == 256 || D == 512) ? 128 : 64;
        auto block_filter = [&](long b0) {
            TopkFilter f = P.filt;
            f.cand += b0 * RANK_SLOTS * P.cap; f.thr_key += b0; f.count += b0 * RANK_SLOTS;
            if (f.exclude) f.exclude += b0;
            return f;
        };
        StageTimer st(c, s);
        for (long b0 = 0; b0 < m; b0 += QBLK)
            HIP_TRY(launch_sweep_bf16(q + (o + b0) * D, gallery, P.sample + b0 * P.ld, P.ld, (int)std::min<long>(QBLK, m - b0), N, D, P.S, P.R,
                                      nullptr, nullptr, s));
        HIP_TRY(launch_topk_sample_bound(P.sample, P.ld, m, P.S, P.R, K, ex, idx_offset, P.thr, P.count, P.flags, P.state, s));
        st.sweep_begin();
        for (long b0 = 0; b0 < m; b0 += QBLK) {
            const int mb = (int)std::min<long>(QBLK, m - b0);
            const TopkFilter f = block_filter(b0);
            HIP_TRY(launch_sweep_bf16(q + (o + b0) * D, gallery, nullptr, 0, mb, N, D, 0, 1, &f, nullptr, s));
            st.sweep_end((double)N * D * 2 + (double)mb * D * 4 + (double)mb * K * 8);
        }
        HIP_TRY(launch_topk_candidates(P.filt, m, K, idx_offset, out_scores + o * K, out_idx + o * K, P.flags, P.state, s));
        HIP_TRY(launch_rank_exact(q + o * D, gallery, 1, m, N, D, K, P.state, P.thr, ex, idx_offset, idx_offset, P.partial, P.groups, P.done,
                                  out_scores + o * K, out_idx + o * K, P.flags, s));
        st.commit(m, (int)N, D);
    }
    return FERN_OK;
}

// Certified bf16 pre-filter + exact fp32 rescoring (include/fern.h: fern_sim_topk_prefiltered).  Same plan buffers and kernels as the
// bf16 sweep for steps 1-3; the bound is lowered by the certified margin, and the final kernel rescored the survivors with the exact
// fp32 chain from the fp32 gallery.  Shapes the bf16 sweep does not cover run fern_sim_topk (same results by definition).
extern "C" int fern_gallery_prepare(fern_ctx* c, const float* gallery, int64_t N, int D, uint16_t* out_bf16, float* out_meta, void* stream) {
    if (!c || N < 0 || D <= 0 || !out_meta || (N && (!gallery || !out_bf16))) return fail(FERN_ERR_ARG, "fern_gallery_prepare: bad argument");
    if (D % 4) return fail(FERN_ERR_ARG, "fern_gallery_prepare: D must be a multiple of 4");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_gallery_prepare(gallery, out_bf16, (long)N, D, out_meta, (hipStream_t)stream));
    return FERN_OK;
}

// The bf16 sweep's score matrix and tile maxima on their own (include/fern.h): what the dense form of fern_sim_topk_prefiltered selects on.
extern "C" int fern_sweep_bf16_scores(fern_ctx* c, const float* q, const uint16_t* gallery_bf16, int B, int64_t N, int D, float* scores, int64_t ld,
                                      float* tile_max, int64_t ldt, void* stream) {
    if (!c || B < 0 || N < 0 || D <= 0 || D % 64 || D > 768) return fail(FERN_ERR_ARG, "fern_sweep_bf16_scores: need D % 64 == 0, D <= 768");
    if ((long)B * N && (!q || !gallery_bf16 || !scores || ld < N || (tile_max && ldt < (N + 31) / 32)))
        return fail(FERN_ERR_ARG, "fern_sweep_bf16_scores: NULL argument or leading dimension too small");
    HIP_TRY(hipSetDevice(c->device));
    for (long b0 = 0; b0 < B; b0 += 64)
        HIP_TRY(launch_sweep_bf16(q + b0 * D, gallery_bf16, scores + b0 * ld, ld, (int)std::min<long>(64, B - b0), N, D, N, 1, nullptr, nullptr,
                                  (hipStream_t)stream, nullptr, tile_max ? tile_max + b0 * ldt : nullptr, ldt));
    return FERN_OK;
}

static long prefilter_sample_rows(int64_t N) {
    // a denser sample than the plain plan's (1 in 32 instead of 1 in 64 beyond 131k rows): the margin lowers the bound by ~0.1 sigma of the
    // score distribution, a tighter sample bound pays that back; 32768 is what the bound kernel holds in registers
    return std::min<long>(std::max<long>(N / 32, std::min<long>(N, 4096)), 32768);
}

// Which form of the stage runs (results are identical, bit for bit): FERN_RANK_PLAIN = fern_sim_topk's fp32-MFMA sweep;
// FERN_RANK_LISTS = bf16 sample pass + bound - margin + filtered bf16 sweep into candidate lists + rescoring (five launches; for
// galleries whose [B, N] score matrix would be real traffic); FERN_RANK_DENSE = bf16 sweep that stores its scores + one
// select-and-rescore kernel (three launches; small galleries, where the stage is launch boundaries, not bytes).
static int rank_strategy_for(const fern_ctx* c, int B, int64_t N, int D) {
    const bool dense_ok = (double)std::min<long>(B, (long)kRankQueryChunk) * N * 4 <= 1.1e9;      // the [B, N] fp32 score matrix of a query chunk: workspace
    if (c->rank_strategy != FERN_RANK_AUTO) return (c->rank_strategy == FERN_RANK_DENSE && !dense_ok) ? (int)FERN_RANK_LISTS : c->rank_strategy;
    // microseconds, fitted to tools/rank_bench.py on MI355X (profiles/r05_rank_bench.txt): one bf16 pass streams at ~5.8 TB/s behind ~14 us of
    // ramp; the dense sweep also writes its [B, N] scores (~4 TB/s); its select kernel reads N / 32 tile maxima per query when the sweep left
    // them (one 64-query block per launch, >= 16 384 rows: fern_sim_topk_prefiltered), else walks the row twice with ONE workgroup (~60 GB/s)
    const long QBLK2 = (D == 64 || D == 128 || D == 256 || D == 512) ? 128 : 64;
    const bool tiles = N >= 16384 && (B <= 64 || QBLK2 == 64 || N >= 131072);
    const long qblk = tiles ? 64 : QBLK2;
    const double sweep_us = std::max(17.0, (double)N * D * 2 / 5.8e6 + 14.0), rounds = (double)((B + 255) / 256);
    const double plain = 38.0 + 2.0 * B * (double)N * D / 95e6;                                        // launches + fp32 MFMA at ~95 TFLOP/s (skinny M)
    const double dense = (double)((B + qblk - 1) / qblk) * (sweep_us + (double)std::min<long>(B, qblk) * N * 4 / 4e6) +
                         rounds * (tiles ? 22.0 + (double)N * 3e-5 : 14.0 + (double)N * 8 / 6e4) + 5.0;
    const double lists = (double)((B + QBLK2 - 1) / QBLK2) * (sweep_us + 25.0) + 16.0 * rounds + 45.0;
    int best = FERN_RANK_PLAIN;
    double t = plain;
    if (lists < t) { best = FERN_RANK_LISTS; t = lists; }
    if (dense_ok && dense < t) best = FERN_RANK_DENSE;
    return best;
}

extern "C" int fern_rank_set_strategy(fern_ctx* c, int strategy) {
    if (!c || strategy < FERN_RANK_AUTO || strategy > FERN_RANK_DENSE) return fail(FERN_ERR_ARG, "fern_rank_set_strategy: unknown strategy");
    c->rank_strategy = strategy;
    return FERN_OK;
}

extern "C" int fern_sim_topk_prefiltered(fern_ctx* c, const float* q, const float* gallery, const uint16_t* gallery_bf16, const float* meta, int B,
                                         int64_t N, int D, int K, float* out_scores, int32_t* out_idx, int64_t idx_offset,
                                         const int32_t* exclude_idx, void* stream) {
    if (!c) return fail(FERN_ERR_ARG, "fern_sim_topk_prefiltered: ctx is NULL");
    if (B < 0 || N < 0 || K < 1 || K > 64 || D <= 0 || D % 32) return fail(FERN_ERR_ARG, "fern_sim_topk_prefiltered: need 1<=K<=64, D % 32 == 0");
    if (B && (!q || !out_scores || !out_idx || (N && (!gallery || !gallery_bf16 || !meta)))) return fail(FERN_ERR_ARG, "fern_sim_topk_prefiltered: NULL argument");
    if (N > 0x7FFFFFF0LL) return fail(FERN_ERR_ARG, "fern_sim_topk_prefiltered: N too large for int32 indices");
    const int strategy = (D % 64 || D > 768 || N == 0 || B == 0) ? (int)FERN_RANK_PLAIN : rank_strategy_for(c, B, N, D);
    if (strategy == FERN_RANK_PLAIN)      // (also: outside the bf16 sweep's shapes -- its LDS ring needs >= 3 stages)
        return fern_sim_topk(c, q, gallery, B, N, D, K, out_scores, out_idx, idx_offset, exclude_idx, stream);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const long QBLK2 = (D == 64 || D == 128 || D == 256 || D == 512) ? 128 : 64, QBLK = QBLK2;
    const double alg_bytes_per_q = (double)D * 4 + (double)K * 8;
    for (long o = 0; o < B; o += (long)kRankQueryChunk) {
        const int m = (int)std::min<long>((long)kRankQueryChunk, B - o);
        const int32_t* ex = exclude_idx ? exclude_idx + o : nullptr;
        FERN_TRY(ws_begin(c, s));
        if (strategy == FERN_RANK_DENSE) {
            const long ld = (N + 31) & ~31L, ldt = (((N + 31) >> 5) + 3) & ~3L;      // whole 32-row tiles; one maximum per tile
            // the sweep leaves tile maxima for the select kernel when it runs one 64-query block per launch; two-block sweeps (65..128
            // queries, D a power of two) have no registers for it: kept below 131 072 rows, where walking the rows costs less than a second
            // gallery pass
            const long QBLK = (m > 64 && N < 131072) ? QBLK2 : 64;
            const bool tiles = QBLK == 64 && N >= 16384;
            const int groups = (int)std::min<long>(256, std::max<long>(1, (N + 4095) / 4096));
            float *approx, *tmax; unsigned long long *thr, *partial; int *flags, *state;
            FERN_TRY(ws_get(c, (size_t)m * ld, &approx));
            FERN_TRY(ws_get(c, (size_t)m * ldt, &tmax));
            FERN_TRY(ws_get(c, (size_t)m, &thr));
            FERN_TRY(ws_get(c, (size_t)4, &flags));
            FERN_TRY(ws_get(c, (size_t)2 * m, &state));
            FERN_TRY(ws_get(c, (size_t)m * groups * 64, &partial));
            StageTimer st(c, s);
            st.sweep_begin();
            for (long b0 = 0; b0 < m; b0 += QBLK) {
                const int mb = (int)std::min<long>(QBLK, m - b0);
                HIP_TRY(launch_sweep_bf16(q + (o + b0) * D, gallery_bf16, approx + b0 * ld, ld, mb, N, D, N, 1, nullptr, nullptr, s,
                                          b0 == 0 ? flags : nullptr, tiles ? tmax + b0 * ldt : nullptr, ldt));
                // bytes this kernel moves: the bf16 copy once, the queries, its [mb, N] fp32 scores out
                st.sweep_end((double)N * D * 2 + (double)mb * D * 4 + (double)mb * N * 4);
            }
            // small galleries: a query without room is ranked by its own workgroup inside the kernel (one CU streams <= 128 MB of fp32 rows:
            // <= ~2 ms in a case that almost never happens) and the gated exact-pass launch -- ~4.5 us of every call -- is not made
            const bool inline_exact = (double)N * D * 4 <= 128e6;
            HIP_TRY(launch_topk_dense_rescore(approx, ld, N, q + o * D, gallery, D, meta, m, K, ex, idx_offset, idx_offset, out_scores + o * K,
                                              out_idx + o * K, thr, flags, state, state + m, s, inline_exact ? 1 : 0, tiles ? tmax : nullptr, ldt));
            if (!inline_exact)
                HIP_TRY(launch_rank_exact(q + o * D, gallery, 0, m, N, D, K, state, thr, ex, idx_offset, idx_offset, partial, groups, state + m,
                                          out_scores + o * K, out_idx + o * K, flags, s));
            st.commit(m, (int)N, D);
            continue;
        }
        RankPlan P;
        FERN_TRY(rank_plan(c, m, N, K, ex, idx_offset, &P, prefilter_sample_rows(N)));
        float* margin;
        FERN_TRY(ws_get(c, (size_t)m, &margin));
        auto block_filter = [&](long b0) {
            TopkFilter f = P.filt;
            f.cand += b0 * RANK_SLOTS * P.cap; f.thr_key += b0; f.count += b0 * RANK_SLOTS;
            if (f.exclude) f.exclude += b0;
            return f;
        };
        StageTimer st(c, s);
        for (long b0 = 0; b0 < m; b0 += QBLK)
            HIP_TRY(launch_sweep_bf16(q + (o + b0) * D, gallery_bf16, P.sample + b0 * P.ld, P.ld, (int)std::min<long>(QBLK, m - b0), N, D, P.S, P.R,
                                      nullptr, nullptr, s));
        HIP_TRY(launch_topk_sample_bound(P.sample, P.ld, m, P.S, P.R, K, ex, idx_offset, P.thr, P.count, P.flags, P.state, s, q + o * D, D, meta, margin));
        st.sweep_begin();
        for (long b0 = 0; b0 < m; b0 += QBLK) {
            const int mb = (int)std::min<long>(QBLK, m - b0);
            const TopkFilter f = block_filter(b0);
            HIP_TRY(launch_sweep_bf16(q + (o + b0) * D, gallery_bf16, nullptr, 0, mb, N, D, 0, 1, &f, nullptr, s));
            // bytes this kernel streams: the bf16 copy once, the queries, nothing stored (the STAGE's algorithmic bytes -- SURVEY 8d, an fp32
            // gallery: N D 4 -- are the caller's to quote against the stage time)
            st.sweep_end((double)N * D * 2 + mb * alg_bytes_per_q);
        }
        HIP_TRY(launch_topk_rescore(P.filt, q + o * D, gallery, D, margin, m, K, idx_offset, out_scores + o * K, out_idx + o * K, P.flags, P.state, s));
        HIP_TRY(launch_rank_exact(q + o * D, gallery, 0, m, N, D, K, P.state, P.thr, ex, idx_offset, idx_offset, P.partial, P.groups, P.done,
                                  out_scores + o * K, out_idx + o * K, P.flags, s));
        st.commit(m, (int)N, D);
    }
    return FERN_OK;
}

extern "C" int fern_gather_scores(fern_ctx* c, const float* q, const float* gallery, const int32_t* idx, float* out, int B, int m, int D,
                                  void* stream) {
    if (!c || B < 0 || m < 0 || D <= 0 || ((long)B * m && (!q || !gallery || !idx || !out))) return fail(FERN_ERR_ARG, "fern_gather_scores: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_gather_scores(q, gallery, idx, out, B, m, D, (hipStream_t)stream));
    return FERN_OK;
}

extern "C" int fern_topk_merge(fern_ctx* c, const float* scores, const int32_t* idx, float* out_scores, int32_t* out_idx, int R, int B, int K,
                               void* stream) {
    if (!c || R < 1 || B < 0 || K < 1 || K > 64 || (B && (!scores || !idx || !out_scores || !out_idx))) return fail(FERN_ERR_ARG, "fern_topk_merge: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_topk_merge(scores, idx, out_scores, out_idx, R, B, K, (hipStream_t)stream));
    return FERN_OK;
}

// ------------------------------------------------------------------------------------------------
// image side: PIL-exact 8-bit resampling + ToTensor/Normalize
// ------------------------------------------------------------------------------------------------
extern "C" int fern_resample_u8_horizontal(fern_ctx* c, const uint8_t* src, int64_t src_ld, int x0, int y0, int rows, uint8_t* dst, int ow,
                                           const int32_t* bounds, const int32_t* coeffs, int ksize, void* stream) {
    if (!c || !src || !dst || !bounds || !coeffs || rows < 0 || ow < 0 || ksize < 1) return fail(FERN_ERR_ARG, "fern_resample_u8_horizontal: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_resample_h(src, src_ld, x0, y0, rows, dst, ow, bounds, coeffs, ksize, (hipStream_t)stream));
    return FERN_OK;
}
extern "C" int fern_resample_u8_vertical(fern_ctx* c, const uint8_t* src, int64_t src_ld, int x0, int y0, int cols, uint8_t* dst, int oh,
                                         const int32_t* bounds, const int32_t* coeffs, int ksize, void* stream) {
    if (!c || !src || !dst || !bounds || !coeffs || cols < 0 || oh < 0 || ksize < 1) return fail(FERN_ERR_ARG, "fern_resample_u8_vertical: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_resample_v(src, src_ld, x0, y0, cols, dst, oh, bounds, coeffs, ksize, (hipStream_t)stream));
    return FERN_OK;
}
extern "C" int fern_u8_to_normalized_chw(fern_ctx* c, const uint8_t* src, int64_t src_ld, int x0, int y0, int64_t src_image_stride, float* dst,
                                         int n, int oh, int ow, const float* host_mean, const float* host_std, void* stream) {
    if (!c || !src || !dst || !host_mean || !host_std || n < 0 || oh < 0 || ow < 0) return fail(FERN_ERR_ARG, "fern_u8_to_normalized_chw: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_u8_to_chw(src, src_ld, x0, y0, dst, n, src_image_stride, oh, ow, host_mean, host_std, (hipStream_t)stream));
    return FERN_OK;
}

// ------------------------------------------------------------------------------------------------
// building blocks
// ------------------------------------------------------------------------------------------------
extern "C" int fern_gemm(fern_ctx* c, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* residual,
                         float* C, int64_t ldc, int M, int N, int K, int epilogue, void* stream) {
    if (!c || M < 0 || N < 0 || K <= 0) return fail(FERN_ERR_ARG, "fern_gemm: bad argument");
    if (M == 0 || N == 0) return FERN_OK;
    if (!A || !W || !C) return fail(FERN_ERR_ARG, "fern_gemm: NULL argument");
    if (epilogue < FERN_EPI_BIAS || epilogue > FERN_EPI_BIAS_RESIDUAL) return fail(FERN_ERR_ARG, "fern_gemm: unknown epilogue");
    if (epilogue == FERN_EPI_BIAS_RESIDUAL && !residual) return fail(FERN_ERR_ARG, "fern_gemm: residual is NULL");
    if (K % 32) return fail(FERN_ERR_ARG, "fern_gemm: K must be a multiple of 32");
    HIP_TRY(hipSetDevice(c->device));
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.R = residual; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.aload = ALOAD_PLAIN;
    return run_gemm(c, p, (hipStream_t)stream);
}

extern "C" int fern_gemm_bf16(fern_ctx* c, const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias,
                              const float* residual, void* C, int64_t ldc, int M, int N, int K, int epilogue, int out_bf16, void* stream) {
    if (!c || M < 0 || N < 0 || K <= 0) return fail(FERN_ERR_ARG, "fern_gemm_bf16: bad argument");
    if (M == 0 || N == 0) return FERN_OK;
    if (!A || !W || !C) return fail(FERN_ERR_ARG, "fern_gemm_bf16: NULL argument");
    if (epilogue < FERN_EPI_BIAS || epilogue > FERN_EPI_BIAS_RESIDUAL) return fail(FERN_ERR_ARG, "fern_gemm_bf16: unknown epilogue");
    if (epilogue == FERN_EPI_BIAS_RESIDUAL && (!residual || out_bf16)) return fail(FERN_ERR_ARG, "fern_gemm_bf16: the residual epilogue needs a residual and fp32 output");
    if (K % 32 || lda % 8 || ldw % 8) return fail(FERN_ERR_ARG, "fern_gemm_bf16: K % 32, lda % 8 and ldw % 8 must be 0");
    HIP_TRY(hipSetDevice(c->device));
    GemmParams p{};
    p.Ab = A; p.lda = lda; p.Wb = W; p.ldw = ldw; p.bias = bias; p.R = residual; p.C = reinterpret_cast<float*>(C); p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.aload = ALOAD_PLAIN; p.out_bf16 = out_bf16 ? 1 : 0;
    int slot;
    FERN_TRY(prof_open(c, PROF_GEMM, 2.0 * M * (double)N * K, (hipStream_t)stream, &slot, M, N, K, 100 + epilogue));
    const hipError_t le = launch_gemm_bf16(p, (hipStream_t)stream);
    HIP_TRY_PROF(le, c, slot);
    return prof_close(c, slot, (hipStream_t)stream);
}

extern "C" int fern_quantize_rows_fp8(fern_ctx* c, const void* x, int x_is_bf16, int64_t ldx, uint8_t* y, int64_t ldy, float* scale, int64_t rows,
                                      int d, void* stream) {
    if (!c || rows < 0 || (rows && (!x || !y || !scale))) return fail(FERN_ERR_ARG, "fern_quantize_rows_fp8: bad argument");
    if (d <= 0 || d % 8 || d > 4096 || ldx % 8 || ldy % 8) return fail(FERN_ERR_ARG, "fern_quantize_rows_fp8: need d % 8 == 0, d <= 4096, ld % 8 == 0");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_quantize_rows_fp8(x_is_bf16 ? static_cast<const unsigned short*>(x) : nullptr, x_is_bf16 ? nullptr : static_cast<const float*>(x),
                                     ldx, y, ldy, scale, rows, d, (hipStream_t)stream));
    return FERN_OK;
}

extern "C" int fern_gemm_fp8(fern_ctx* c, const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, int64_t ldw, const float* scale_w,
                             const float* bias, const float* residual, void* C, int64_t ldc, int M, int N, int K, int epilogue, int out_bf16,
                             void* stream) {
    if (!c || M < 0 || N < 0 || K <= 0) return fail(FERN_ERR_ARG, "fern_gemm_fp8: bad argument");
    if (M == 0 || N == 0) return FERN_OK;
    if (!A || !W || !C || !scale_a || !scale_w) return fail(FERN_ERR_ARG, "fern_gemm_fp8: NULL argument");
    if (epilogue != FERN_EPI_BIAS && epilogue != FERN_EPI_BIAS_GELU && epilogue != FERN_EPI_BIAS_RESIDUAL)
        return fail(FERN_ERR_ARG, "fern_gemm_fp8: epilogue must be BIAS, BIAS_GELU or BIAS_RESIDUAL");
    if (epilogue == FERN_EPI_BIAS_RESIDUAL && (!residual || out_bf16)) return fail(FERN_ERR_ARG, "fern_gemm_fp8: the residual epilogue needs a residual and fp32 output");
    if (K % 64 || lda % 16 || ldw % 16) return fail(FERN_ERR_ARG, "fern_gemm_fp8: K % 64, lda % 16 and ldw % 16 must be 0");
    HIP_TRY(hipSetDevice(c->device));
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A); p.lda = lda; p.Wb = reinterpret_cast<const unsigned short*>(W); p.ldw = ldw;
    p.bias = bias; p.R = residual; p.C = reinterpret_cast<float*>(C); p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    p.aload = ALOAD_PLAIN; p.out_bf16 = out_bf16 ? 1 : 0; p.fp8 = 1; p.scale_a = scale_a; p.scale_w = scale_w;
    return run_gemm_b(c, p, (hipStream_t)stream);
}

extern "C" int fern_quantize_mx8(fern_ctx* c, const void* x, int x_is_bf16, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* scales,
                                 int64_t scale_rows, int64_t rows, int d, void* stream) {
    if (!c || rows < 0 || (rows && (!x || !y || !scales))) return fail(FERN_ERR_ARG, "fern_quantize_mx8: bad argument");
    if (d <= 0 || d % 128 || d > 4096 || ldx % 8 || ldy % 8 || scale_rows < rows)
        return fail(FERN_ERR_ARG, "fern_quantize_mx8: need d % 128 == 0, d <= 4096, ld % 8 == 0, scale_rows >= rows");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_quantize_mx8(x_is_bf16 ? static_cast<const unsigned short*>(x) : nullptr, x_is_bf16 ? nullptr : static_cast<const float*>(x), ldx,
                                y, ldy, scales, scale_rows, rows, d, (hipStream_t)stream));
    return FERN_OK;
}

extern "C" int fern_gemm_mx8(fern_ctx* c, const uint8_t* A, int64_t lda, const uint8_t* scales_a, int64_t scale_rows_a, const uint8_t* W,
                             int64_t ldw, const uint8_t* scales_w, int64_t scale_rows_w, const float* bias, const float* residual, void* C,
                             int64_t ldc, int M, int N, int K, int epilogue, int out_bf16, void* stream) {
    if (!c || M < 0 || N < 0 || K <= 0) return fail(FERN_ERR_ARG, "fern_gemm_mx8: bad argument");
    if (M == 0 || N == 0) return FERN_OK;
    if (!A || !W || !C || !scales_a || !scales_w) return fail(FERN_ERR_ARG, "fern_gemm_mx8: NULL argument");
    if (epilogue != FERN_EPI_BIAS && epilogue != FERN_EPI_BIAS_GELU && epilogue != FERN_EPI_BIAS_RESIDUAL)
        return fail(FERN_ERR_ARG, "fern_gemm_mx8: epilogue must be BIAS, BIAS_GELU or BIAS_RESIDUAL");
    if (epilogue == FERN_EPI_BIAS_RESIDUAL && !residual) return fail(FERN_ERR_ARG, "fern_gemm_mx8: the residual epilogue needs a residual");
    if (K % 128 || lda % 16 || ldw % 16 || scale_rows_a < M || scale_rows_w < N)
        return fail(FERN_ERR_ARG, "fern_gemm_mx8: K % 128, lda % 16 and ldw % 16 must be 0, scale_rows >= rows");
    HIP_TRY(hipSetDevice(c->device));
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A); p.lda = lda; p.Wb = reinterpret_cast<const unsigned short*>(W); p.ldw = ldw;
    p.bias = bias; p.R = residual; p.C = reinterpret_cast<float*>(C); p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    p.aload = ALOAD_PLAIN; p.out_bf16 = out_bf16 ? 1 : 0; p.fp8 = 2; p.mxa = scales_a; p.mxa_rows = scale_rows_a; p.mxw = scales_w; p.mxw_rows = scale_rows_w;
    if (out_bf16 && epilogue == FERN_EPI_BIAS_RESIDUAL) { p.Rb = reinterpret_cast<const unsigned short*>(residual); p.R = nullptr; }      // bf16 residual stream
    return run_gemm_b(c, p, (hipStream_t)stream);
}

extern "C" int fern_gemm_mx8_quant(fern_ctx* c, const uint8_t* A, int64_t lda, const uint8_t* scales_a, int64_t scale_rows_a, const uint8_t* W,
                                   int64_t ldw, const uint8_t* scales_w, int64_t scale_rows_w, const float* bias, uint8_t* C8, int64_t ldc,
                                   uint8_t* scales_c, int64_t scale_rows_c, int M, int N, int K, int epilogue, void* stream) {
    if (!c || M < 0 || N < 0 || K <= 0) return fail(FERN_ERR_ARG, "fern_gemm_mx8_quant: bad argument");
    if (M == 0 || N == 0) return FERN_OK;
    if (!A || !W || !C8 || !scales_a || !scales_w || !scales_c) return fail(FERN_ERR_ARG, "fern_gemm_mx8_quant: NULL argument");
    if (epilogue != FERN_EPI_BIAS && epilogue != FERN_EPI_BIAS_GELU) return fail(FERN_ERR_ARG, "fern_gemm_mx8_quant: epilogue must be BIAS or BIAS_GELU");
    if (K % 128 || N % 128 || lda % 16 || ldw % 16 || ldc % 16 || scale_rows_a < M || scale_rows_w < N || scale_rows_c < M)
        return fail(FERN_ERR_ARG, "fern_gemm_mx8_quant: K % 128, N % 128, lda / ldw / ldc % 16 must be 0, scale_rows >= rows");
    HIP_TRY(hipSetDevice(c->device));
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A); p.lda = lda; p.Wb = reinterpret_cast<const unsigned short*>(W); p.ldw = ldw;
    p.bias = bias; p.C = reinterpret_cast<float*>(C8); p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    p.aload = ALOAD_PLAIN; p.fp8 = 2; p.mxa = scales_a; p.mxa_rows = scale_rows_a; p.mxw = scales_w; p.mxw_rows = scale_rows_w;
    p.out_mx8 = 1; p.mxc = scales_c; p.mxc_rows = scale_rows_c;
    return run_gemm_b(c, p, (hipStream_t)stream);
}

extern "C" int fern_layernorm(fern_ctx* c, const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                              int64_t rows, int d, float eps, void* stream) {
    if (!c || !x || !gamma || !beta || !y || rows < 0) return fail(FERN_ERR_ARG, "fern_layernorm: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_layernorm(x, residual, gamma, beta, y, rows, d, d, d, eps, (hipStream_t)stream));
    return FERN_OK;
}

extern "C" int fern_attention(fern_ctx* c, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out,
                              int64_t ldo, int batch, int heads, int head_dim, int s_q, int s_k, int causal, float scale, void* stream) {
    if (!c || !q || !k || !v || !out) return fail(FERN_ERR_ARG, "fern_attention: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    AttnParams a{q, k, v, out, (long)ldq, (long)ldk, (long)ldv, (long)ldo, batch, heads, head_dim, s_q, s_k, causal, scale};
    return run_attention(c, a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// profiling
// ------------------------------------------------------------------------------------------------
extern "C" int fern_attention_bf16(fern_ctx* c, const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                                   uint16_t* out, int64_t ldo, int batch, int heads, int head_dim, int s_q, int s_k, int causal, float scale,
                                   void* stream) {
    if (!c || !q || !k || !v || !out) return fail(FERN_ERR_ARG, "fern_attention_bf16: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    AttnParams a{nullptr, nullptr, nullptr, nullptr, (long)ldq, (long)ldk, (long)ldv, (long)ldo, batch, heads, head_dim, s_q, s_k, causal, scale,
                 out, q, k, v};
    return run_attention(c, a, (hipStream_t)stream);
}

// The tuner's per-shape tile choices of this process (every context shares them), as text: one line per shape,
// "f32|bf16|fp8 M N K epilogue loader|outflags cfg".  Returns the number of bytes the full text needs (excluding the
// terminator); writes at most cap - 1 bytes + NUL.  A file of these lines, named by FERN_GEMM_TILES, pins the choices.
extern "C" int64_t fern_tuner_export(char* buf, int64_t cap) {
    std::string text;
    gemm_tuner_export(text);
    gemm_bf16_tuner_export(text);
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(text.size(), (size_t)cap - 1);
        std::memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return (int64_t)text.size();
}

extern "C" uint64_t fern_ws_generation(const fern_ctx* c) { return c ? c->ws_generation : 0; }

// The reverse of fern_tuner_export: `text` (NUL-terminated lines of the export format) replaces this process's choices for the
// shapes it lists -- e.g. rank 0's export broadcast to every rank, so that all ranks of a job run identical kernels.  Unknown or
// inapplicable lines are skipped.  Every configuration is bit-identical, so this never changes a result.
extern "C" int fern_tuner_import(const char* text) {
    if (!text) return fail(FERN_ERR_ARG, "fern_tuner_import: text is NULL");
    const std::string t(text);
    gemm_tuner_import(t);
    gemm_bf16_tuner_import(t);
    return FERN_OK;
}

// How many batches the caller keeps in flight on separate streams (process-wide, like the tile choices themselves): the
// reduced-precision GEMM families then score a trial by duration x (share of the chip it fills)^0.75 instead of duration alone.
// Set before the first launch of a shape; shapes already tuned keep their choice.  Never changes a result.
extern "C" int fern_tuner_set_concurrency(int lanes) {
    if (lanes < 1) return fail(FERN_ERR_ARG, "fern_tuner_set_concurrency: lanes must be >= 1");
    gemm_bf16_tuner_set_concurrency(lanes);
    return FERN_OK;
}

// Force ONE tile configuration of a GEMM family, process-wide, until released (cfg < 0): what FERN_GEMM_CFG / FERN_GEMM_SPLIT_CFG /
// FERN_GEMM_BF16_CFG / FERN_GEMM_FP8_CFG / FERN_GEMM_MX8_CFG do from the environment, switchable at run time so that a test process can walk
// every variant.  A configuration that cannot serve a call (k tile does not divide K, ...) falls back to the family's own choice.
extern "C" int fern_tuner_force_config(const char* family, int cfg) {
    static const char* names[] = {"f32", "f32x3", "bf16", "fp8", "mx8"};
    for (int f = 0; family && f < 5; ++f)
        if (!std::strcmp(family, names[f])) return (f < 2 ? gemm_force_cfg(f, cfg) : gemm_bf16_force_cfg(f, cfg)) ? FERN_OK : fail(FERN_ERR_ARG, "fern_tuner_force_config");
    return fail(FERN_ERR_ARG, "fern_tuner_force_config: family is one of f32, f32x3, bf16, fp8, mx8");
}

extern "C" int fern_prof_enable(fern_ctx* c, int on) {
    if (!c) return fail(FERN_ERR_ARG, "fern_prof_enable: ctx is NULL");
    c->prof_on = on != 0;
    return FERN_OK;
}
extern "C" int fern_prof_collect(fern_ctx* c, fern_prof_stats* out) {
    if (!c || !out) return fail(FERN_ERR_ARG, "fern_prof_collect: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    std::memset(out, 0, sizeof(*out));
    // FERN_PROF_DUMP=<path>: append one CSV line per instrumented launch (kind,m,n,k,tag,ms,work) for shape-level analysis
    double stage_ms = 0.0;
    const char* dump_path = std::getenv("FERN_PROF_DUMP");
    FILE* dump = dump_path ? std::fopen(dump_path, "a") : nullptr;
    for (auto& r : c->recs) {
        float ms = 0.f;
        if (r.span) {                    // the ranking stage: first dispatch begin -> last dispatch end; its sweep dispatches on their own
            HIP_TRY(hipEventElapsedTime(&ms, r.kev.front(), r.kev.back()));
            for (int i = r.sweep_lo; i < r.sweep_hi; ++i) {
                float one = 0.f;
                HIP_TRY(hipEventElapsedTime(&one, r.kev[2 * i], r.kev[2 * i + 1]));
                out->sweep_ms += one;
                out->sweep_launches++;
            }
            out->sweep_bytes += r.work;
            for (hipEvent_t e : r.kev) c->ev_pool.push_back(e);
            r.kev.clear();
        } else if (!r.a) {               // kernel-precise: the sum of the dispatches' own durations
            for (size_t i = 0; i + 1 < r.kev.size(); i += 2) {
                float one = 0.f;
                HIP_TRY(hipEventElapsedTime(&one, r.kev[i], r.kev[i + 1]));
                ms += one;
            }
            for (hipEvent_t e : r.kev) c->ev_pool.push_back(e);
            r.kev.clear();
        } else
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        if (dump) std::fprintf(dump, "%d,%d,%d,%d,%d,%.6f,%.0f\n", r.kind, r.m, r.n, r.k, r.tag, ms, r.work);
        switch (r.kind) {
            case PROF_GEMM:
                if (r.tag >= 300) { out->gemm_mx8_ms += ms; out->gemm_mx8_flops += r.work; out->gemm_mx8_launches++; out->gemm_mx8_bf16_flops += r.extra_flops; }
                else if (r.tag >= 200) { out->gemm_fp8_ms += ms; out->gemm_fp8_flops += r.work; out->gemm_fp8_launches++; }
                else if (r.tag >= 100) { out->gemm_bf16_ms += ms; out->gemm_bf16_flops += r.work; out->gemm_bf16_launches++; }
                else { out->gemm_ms += ms; out->gemm_flops += r.work; out->gemm_launches++;
                       out->gemm_alg_bytes += 4.0 * ((double)r.m * r.k + (double)r.n * r.k + (double)r.m * r.n) + r.extra_alg_bytes;
                       out->gemm_dispatches += r.dispatches; }
                break;
            case PROF_ATTN: out->attn_ms += ms; out->attn_flops += r.work; out->attn_launches++; break;
            case PROF_TOPK: out->topk_ms += ms; out->topk_launches++; break;
            case PROF_STAGE: stage_ms += ms; out->topk_launches++; break;      // whole ranking stage of a query chunk (a span record, or one marker pair)
            default: out->sweep_ms += ms; out->sweep_bytes += r.work; out->sweep_launches++; break;
        }
        if (r.a) { c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b); }
    }
    c->recs.clear();
    // topk_ms = the rest of the ranking stage = (stage intervals, launch boundaries included) - (the sweep kernels' own durations)
    if (stage_ms > 0.0) out->topk_ms += stage_ms > out->sweep_ms ? stage_ms - out->sweep_ms : 0.0;
    if (dump) std::fclose(dump);
    return FERN_OK;
}
