// HBM-bound cosine sweep over a bf16 gallery (BASELINE config 5: "bf16 similarity"): scores[q][n] = Q[q] . G[n] with
// fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// In fp32 the B = 64 sweep is MFMA-bound (2*B*N*D / 157 TF > N*D*4 / 8 TB/s); with a bf16 gallery the matrix cores are
// 16x faster and the gallery bytes halve, so the kernel is a pure HBM stream and is built as one:
//   * the (<= 64) queries are converted to bf16 once and stay in LDS (padded rows: conflict-free ds_read_b128);
//   * every WAVE streams its own 32-row gallery tiles through a private LDS ring with global_load_lds_dwordx4
//     (4 KiB stages = 32 rows x 64 k, swizzle on the per-lane source address) -- there is NO workgroup barrier in the
//     main loop, only the wave's own counted s_waitcnt vmcnt(N), and STAGES-1 stages (12-16 KiB per wave, 48-64 KiB
//     per CU) stay in flight to cover HBM latency;
//   * MFMA operands: A = queries (rows on the registers), B = gallery rows (row on the lane), so a 32x32 accumulator
//     register holds 32 consecutive gallery rows of one query: score stores are 128-byte coalesced.
// The [B, N] score matrix is never stored: a small SAMPLE pass (every R-th row, jittered) is stored and yields a per-query
// lower bound of the K-th best score; the full pass compares each finished 32x32 score tile with the bounds in registers and
// appends only the survivors (~K*R of N per query) to per-query candidate lists (kernels.h: TopkFilter, topk.hip).
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace fern {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f32_to_bf16_rne(float f) { return f32_to_bf16_bits(f); }      // round to nearest even (kernels.h)

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* x, u16* y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    ushort4 o;
    o.x = f32_to_bf16_rne(v[0]); o.y = f32_to_bf16_rne(v[1]); o.z = f32_to_bf16_rne(v[2]); o.w = f32_to_bf16_rne(v[3]);
    reinterpret_cast<ushort4*>(y)[i] = o;
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const u16* x, float* y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const uint2 w = reinterpret_cast<const uint2*>(x)[i];
    const f32x4 o = {__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
    reinterpret_cast<f32x4*>(y)[i] = o;
}

// fern_gallery_prepare: the bf16 copy of an fp32 gallery (round to nearest even, as f32_to_bf16_kernel) AND what certifies it as a
// pre-filter of the exact fp32 ranking (api.hip: fern_sim_topk_prefiltered): meta[0] = max_n ||g_n - bf16(g_n)||, meta[1] = max_n
// ||bf16(g_n)||, meta[2] = max_n ||g_n|| (2-norms, fp32; the row sums are added in a fixed lane order, so a row's value -- and so
// the maxima -- do not depend on the launch).  g - bf16(g) is exact in fp32.  One wave per row, D % 4 == 0, any D.
__global__ __launch_bounds__(256) void gallery_prepare_kernel(const float* x, u16* y, long n, int d, float* meta) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float e2 = 0.f, t2 = 0.f, g2 = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * d + c);
        ushort4 o;
        o.x = f32_to_bf16_rne(v[0]); o.y = f32_to_bf16_rne(v[1]); o.z = f32_to_bf16_rne(v[2]); o.w = f32_to_bf16_rne(v[3]);
        *reinterpret_cast<ushort4*>(y + row * d + c) = o;
        const float r0 = bf16_bits_to_f32(o.x), r1 = bf16_bits_to_f32(o.y), r2 = bf16_bits_to_f32(o.z), r3 = bf16_bits_to_f32(o.w);
        const float d0 = v[0] - r0, d1 = v[1] - r1, d2 = v[2] - r2, d3 = v[3] - r3;
        e2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        t2 += r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
        g2 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { e2 += __shfl_xor(e2, m); t2 += __shfl_xor(t2, m); g2 += __shfl_xor(g2, m); }
    if (lane == 0) {      // non-negative floats order like their bit patterns; NaN / inf rows poison the bound upwards (a NaN margin accepts every row)
        atomicMax(reinterpret_cast<unsigned*>(meta + 0), __float_as_uint(sqrtf(e2)));
        atomicMax(reinterpret_cast<unsigned*>(meta + 1), __float_as_uint(sqrtf(t2)));
        atomicMax(reinterpret_cast<unsigned*>(meta + 2), __float_as_uint(sqrtf(g2)));
    }
}

constexpr int ROWS_T = 32;              // gallery rows per wave tile
constexpr int KSTAGE = 64;              // k elements per ring stage (128 bytes per row)
constexpr int STAGE_BYTES = ROWS_T * KSTAGE * 2;   // 4096
// Filter form: survivors are queued per wave in LDS and flushed to the candidate lists QFLUSH.. at a time -- the list append is
// a RETURNING global atomic (~1-2 us round trip); one per finished tile stalled the wave's DMA ring and cost 40 % of the stream
constexpr int QCAP = 128;                          // queue entries per wave (a register can add up to 64 at once)
constexpr int QUEUE_BYTES = 4 * QCAP * 12 + 128 * 8 + 128 * 4;     // 4 waves x (key 8 B + query 4 B) + the (<= 128) bounds as keys and as floats

// FILTER = false: the sample pass -- tile rows are the gallery rows sample_row(c, R) of S sample columns, scores are stored
// ([B, ld], 128-byte coalesced).  FILTER = true: the full sweep -- nothing is stored; every finished 32x32 score tile is
// compared with the per-query bounds held in registers and only survivors are appended to the candidate lists.
// KCH > 0 (D = 64 * KCH known at compile time): the queries' MFMA fragments live in REGISTERS (2 x 4 KCH fragments of 4 VGPRs: 256
// VGPRs at D = 512 -- the kernel runs one wave per SIMD, so it owns the whole 512-entry file), and the LDS the query image
// occupied goes to the ring: 9 stages per wave, 128 KiB in flight per CU instead of 64.  HBM latency under this load is ~4 us,
// and 64 KiB in flight per CU is then 16 GB/s per CU (3.8 TB/s on the chip) by Little's law.  KCH = 0: queries stay in LDS (any D).
// QB = 2 (KCH > 0 only): 65..128 queries in ONE pass over the gallery (round 2 swept the gallery once per 64-query block).  Queries
// 0..63 keep their fragments in registers as before; the bf16 image of queries 64..127 stays in LDS (66 KB at D = 512) and their
// fragments are read per k step -- 3 LDS reads per k step instead of 1, a fifth of the LDS bandwidth at the gallery's HBM rate -- and
// the ring shrinks from 9 to 5 stages per wave to make room.  MFMA time per gallery byte doubles and stays far below the HBM time.
// Lane exchange at distance STEP inside a 32-lane half (the lanes that hold one query's 32 gallery rows): DPP where the pattern exists
// (quad permutes, row rotate by 8), ds_swizzle (no LDS memory touched) for 4 and 16.
template <int STEP>
__device__ __forceinline__ float lane_xchg(float x) {
    int v = __float_as_int(x);
    if (STEP == 1) v = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);             // quad_perm [1, 0, 3, 2]
    else if (STEP == 2) v = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);        // quad_perm [2, 3, 0, 1]
    else if (STEP == 8) v = __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);       // row_ror:8
    else if (STEP == 4) v = __builtin_amdgcn_ds_swizzle(v, 0x101F);                          // bit mode: and 0x1f, xor 0x04
    else v = __builtin_amdgcn_ds_swizzle(v, 0x401F);                                         // bit mode: and 0x1f, xor 0x10
    return __int_as_float(v);
}
// The transposing reduction of a wave tile's scores (A = queries 0..31 of the block, Bv = 32..63, laid out like the MFMA accumulators):
// in each step lanes l and l ^ STEP keep one half of their values each and receive the partner's copy of that half -- after the steps
// (16 values, distance 1) (8, 2) (4, 8) (2, 4) (1, 16) every lane holds the maximum over the 32 lanes (= gallery rows) of ONE value: 31
// exchanges instead of 32 x 5.  Ext-vector values with constant indices only: arrays indexed by the lane's half went to scratch.
typedef float f32x8r __attribute__((ext_vector_type(8)));
typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef float f32x2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tile_row_max(const f32x16& A, const f32x16& Bv, int lane) {
    const bool k1 = (lane & 1) == 0, k2 = (lane & 2) == 0, k8 = (lane & 8) == 0, k4 = (lane & 4) == 0, k16 = (lane & 16) == 0;
    f32x16 r16;
#pragma unroll
    for (int i = 0; i < 16; ++i) r16[i] = fmaxf(k1 ? A[i] : Bv[i], lane_xchg<1>(k1 ? Bv[i] : A[i]));
    f32x8r r8;
#pragma unroll
    for (int i = 0; i < 8; ++i) r8[i] = fmaxf(k2 ? r16[i] : r16[i + 8], lane_xchg<2>(k2 ? r16[i + 8] : r16[i]));
    f32x4r r4;
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = fmaxf(k8 ? r8[i] : r8[i + 4], lane_xchg<8>(k8 ? r8[i + 4] : r8[i]));
    f32x2r r2;
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = fmaxf(k4 ? r4[i] : r4[i + 2], lane_xchg<4>(k4 ? r4[i + 2] : r4[i]));
    return fmaxf(k16 ? r2[0] : r2[1], lane_xchg<16>(k16 ? r2[1] : r2[0]));
}
// ... and which value that is: bit 4 of its index (A or Bv) is chosen by lane bit 0, bit 3 by lane bit 1, bit 2 by lane bit 3, bit 1 by lane
// bit 2, bit 0 by lane bit 4; value i of lane half lh is query (i >> 4) * 32 + (i & 3) + 8 ((i & 15) >> 2) + 4 lh of the block.
__device__ __forceinline__ int tile_row_max_query(int lane) {
    const int i = ((lane & 1) << 4) | ((lane & 2) << 2) | ((lane & 8) >> 1) | ((lane & 4) >> 1) | ((lane & 16) >> 4);
    return (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2) + 4 * (lane >> 5);
}

template <int STAGES, bool FILTER, int KCH, int QB = 1>
__global__ __launch_bounds__(256) void sweep_bf16_kernel(const float* q, const u16* g, float* scores, long ld, int B, long N, int D, long S, int R,
                                                         TopkFilter filt, const int* gate, int* zero_flags, float* tmax, long ldt) {
    static_assert(QB == 1 || KCH > 0, "two query blocks need the register-resident form");
    if (gate && *gate == 0) return;
    if (!FILTER && zero_flags && blockIdx.x == 0 && threadIdx.x == 0) { zero_flags[0] = 0; zero_flags[1] = 0; }
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int q_stride = D * 2 + 16;                               // bytes; +16 spreads rows over the banks
    // KCH > 0, QB = 1: the ring overlays the query image; QB = 2: the image of queries 64..127 stays, the ring follows it.
    // The ring is SLOT-major (slot s of wave w at (4 s + w) stages) and, in the overlay form, the image sits on the LAST slots: the first
    // EARLY slots of every wave are free from the start, so the gallery's first stages are requested BEFORE the queries are staged
    // (their HBM round trip, ~2-4 us at C2's 33 MB burst, used to begin only after the ~4 us of query staging).
    constexpr bool OVERLAY = KCH > 0 && QB == 1;
    constexpr int QIMG = OVERLAY ? ((64 * (128 * KCH + 16) + 1023) & ~1023) : 0;
    constexpr int EARLY_ROOM = OVERLAY ? (4 * STAGES * STAGE_BYTES + QUEUE_BYTES - QIMG) / (4 * STAGE_BYTES) : 0;
    constexpr int EARLY = !OVERLAY ? STAGES - 1 : EARLY_ROOM > STAGES - 1 ? STAGES - 1 : EARLY_ROOM;      // (ring next to the image: every slot is free)
    unsigned char* img = smem + (OVERLAY ? EARLY * 4 * STAGE_BYTES : 0);
    unsigned char* ring_base = OVERLAY ? smem : smem + ((64 * q_stride + 1023) & ~1023);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- queries -> bf16 -> LDS (rows >= B are zero); block 0 = queries 0..63, block 1 = 64..127 ----
    auto stage_queries = [&](int q0) {
        for (int i = tid; i < 64 * (D / 8); i += 256) {
            const int row = i / (D / 8), c8 = (i % (D / 8)) * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q0 + row < B) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(q + (long)(q0 + row) * D + c8);
                const f32x4 b = *reinterpret_cast<const f32x4*>(q + (long)(q0 + row) * D + c8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (short)f32_to_bf16_rne(a[e]); v[4 + e] = (short)f32_to_bf16_rne(b[e]); }
            }
            *reinterpret_cast<bf16x8*>(img + row * q_stride + c8 * 2) = v;
        }
    };
    constexpr int NFR = KCH > 0 ? 8 * KCH : 1;
    bf16x8 afr[NFR];
    // (the register-resident forms stage their queries further down, behind the first gallery requests)
    if (KCH == 0) {
        stage_queries(0);
        __syncthreads();
    }

    // per-query bounds of the filter, laid out like the accumulator registers (query = tm*32 + (r&3) + 8(r>>2) + 4 lh)
    float bound[QB == 1 ? 2 : 1][16];                                // QB = 2: the bounds are re-read from LDS per tile (registers are full)
    unsigned char* qbase = ring_base + 4 * STAGES * STAGE_BYTES;
    unsigned long long* thr_lds = reinterpret_cast<unsigned long long*>(qbase + 4 * QCAP * 12);
    float* thr_f = reinterpret_cast<float*>(qbase + 4 * QCAP * 12 + 128 * 8);      // QB = 2: the bounds as floats, [128]
    unsigned long long* qkey = reinterpret_cast<unsigned long long*>(qbase + wave * QCAP * 12);
    int* qq = reinterpret_cast<int*>(qbase + wave * QCAP * 12 + QCAP * 8);
    int qlen = 0;                                                  // wave-uniform
    // append the queued survivors to their lists: one batch of returning atomics per <= 64 entries
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the queue writes of every lane have landed
        for (int base = 0; base < qlen; base += 64) {
            const int i = base + lane;
            if (i < qlen) {
                const unsigned long long key = qkey[i];
                const int qi = qq[i];
                const long n = (long)(0xFFFFFFFFu - (unsigned)key);
                const long list = (long)qi * RANK_SLOTS + rank_slot(n);
                const int pos = atomicAdd(&filt.count[list], 1);
                if (pos < filt.cap) filt.cand[list * filt.cap + pos] = key;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // queue reads done before the next appends overwrite it
        qlen = 0;
    };
    // one finished accumulator tile (32 queries x 32 gallery rows): fast reject, else queue the survivors
    auto filter_tile = [&](const f32x16& a, const float (&bd)[16], int tm, long n, bool n_ok) {
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) hits |= (!(a[r] < bd[r]) ? 1u : 0u) << r;
        hits = n_ok ? hits : 0u;
        if (!__any(hits != 0)) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (__any((hits >> r) & 1u)) {                           // wave-uniform
                const int qi = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const unsigned long long key = make_key(a[r], (unsigned)n);
                const bool keep = ((hits >> r) & 1u) && key >= thr_lds[qi] &&
                                  !(filt.exclude && qi < B && (long)filt.exclude[qi] - filt.exclude_off == n);
                const unsigned long long km = __ballot(keep);
                if (km) {
                    if (qlen > QCAP - 64) flush();
                    const int off = qlen + __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0));
                    if (keep) { qkey[off] = key; qq[off] = qi; }
                    qlen += __popcll(km);
                }
            }
        }
    };

    constexpr int SLOT_BYTES = 4 * STAGE_BYTES;                     // slot-major: the four waves' stages of one slot are adjacent
    unsigned char* ring = ring_base + wave * STAGE_BYTES;
    const int kchunks = KCH > 0 ? KCH : D / KSTAGE;                // stages per tile (a compile-time power of two on the register path)
    const long rows_total = FILTER ? N : S;
    const long ntiles = (rows_total + ROWS_T - 1) / ROWS_T;
    const long gw = (long)blockIdx.x * 4 + wave, GW = (long)gridDim.x * 4;
    const int my_tiles = gw < ntiles ? (int)((ntiles - gw + GW - 1) / GW) : 0;
    const int nstages = my_tiles * kchunks;

    // DMA source of this lane inside a stage: piece p = 8 rows x 128 B; lane -> row p*8 + lane/8, position lane%8 holds
    // logical chunk (lane%8) ^ ((row >> 1) & 7).  The stage sequence (tile, k chunk, ring slot) advances by running counters:
    // a 64-bit division per stage (stage index -> tile, chunk) was ~300 scalar instructions in a loop that one wave per SIMD
    // runs alone.  Full tiles of the filtered sweep address their rows as (uniform tile base) + (per-lane 32-bit offset).
    long t_issue = gw;                                             // wave-uniform
    int kc_issue = 0, slot_issue = 0, issued = 0;
    unsigned lane_off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 8 + (lane >> 3);
        lane_off[p] = (unsigned)((row * D + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2);      // bytes; 32 rows x D <= 64 KiB
    }
    auto issue_next = [&]() {
        unsigned char* dst = ring + slot_issue * SLOT_BYTES;
        if ((FILTER || R == 1) && (t_issue + 1) * ROWS_T <= (FILTER ? N : S)) {      // whole tile of consecutive rows (the full sweep; the store-all form S = N, R = 1)
            const unsigned char* base = reinterpret_cast<const unsigned char*>(g) + (t_issue * ROWS_T * D + (long)kc_issue * KSTAGE) * 2;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + lane_off[p]),
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 2);      // aux 2 = nt: every gallery byte is read once
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = p * 8 + (lane >> 3);
                long n = t_issue * ROWS_T + row;
                if (!FILTER) n = sample_row(n < S ? n : S - 1, R);
                n = n < N ? n : N - 1;
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const u16* src = g + n * D + kc_issue * KSTAGE + chunk * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 2);
            }
        }
        if (++kc_issue == kchunks) { kc_issue = 0; t_issue += GW; }
        slot_issue = slot_issue + 1 == STAGES ? 0 : slot_issue + 1;
        ++issued;
    };

    f32x16 acc[2 * QB];
#pragma unroll
    for (int i = 0; i < 2 * QB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    int pre = 0;
    if (KCH > 0) {
        // the queries' loads go out FIRST (vmcnt retires in order: behind the gallery requests they would wait for those too), then
        // the gallery's first EARLY stages, then the queries are rounded into the image and every wave takes its fragments
        constexpr int NQ = KCH > 0 ? 2 * KCH : 1;                   // 64 rows x D / 8 chunks over 256 threads
        f32x4 qa[NQ][2];
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int i = tid + 256 * it, row = i / (8 * KCH), c8 = (i % (8 * KCH)) * 8;
            const float* src = q + (long)(row < B ? row : B - 1) * D + c8;      // unconditional (clamped) loads: a guarded load waits for itself
            qa[it][0] = *reinterpret_cast<const f32x4*>(src);
            qa[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
        for (; pre < EARLY && pre < nstages; ++pre) issue_next();
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int i = tid + 256 * it, row = i / (8 * KCH), c8 = (i % (8 * KCH)) * 8;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = row < B ? (short)f32_to_bf16_rne(qa[it][0][e]) : (short)0;
                v[4 + e] = row < B ? (short)f32_to_bf16_rne(qa[it][1][e]) : (short)0;
            }
            *reinterpret_cast<bf16x8*>(img + row * q_stride + c8 * 2) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int kidx = 0; kidx < 4 * KCH; ++kidx)
                afr[tm * 4 * KCH + kidx] = *reinterpret_cast<const bf16x8*>(img + (tm * 32 + l31) * q_stride + (kidx * 16 + lh * 8) * 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                           // every wave holds its fragments: the image's slots may be filled ...
        if (QB == 2) {                                             // ... or the image overwritten by the second query block, which stays in LDS
            stage_queries(64);
            __syncthreads();
        }
    }
    if (FILTER) {
        if (tid < 64 * QB) thr_lds[tid] = tid < B ? filt.thr_key[tid] : ~0ull;
        if (QB == 2 && tid < 128) thr_f[tid] = filter_bound(tid < B ? filt.thr_key[tid] : ~0ull);
        __syncthreads();
        if (QB == 1) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) bound[tm][r] = filter_bound(thr_lds[tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh]);
        }
    }
    for (; pre < STAGES - 1 && pre < nstages; ++pre) issue_next();
    const int sw = (l31 >> 1) & 7;
    int slot_read = 0;                                             // ring slot of the stage being consumed
    // one ring stage: refill the slot that was just read, wait for this stage (the newer ones stay in flight), 4 k-steps of MFMAs
    auto stage_step = [&](int kc) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot about to be refilled has been read
        if (issued < nstages) {
            issue_next();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (STAGES - 1)) : "memory");   // this stage landed; newer ones stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* st = ring + slot_read * SLOT_BYTES;
        slot_read = slot_read + 1 == STAGES ? 0 : slot_read + 1;
#pragma unroll
        for (int ks = 0; ks < KSTAGE / 16; ++ks) {
            const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(st + l31 * 128 + (((2 * ks + lh) ^ sw) * 16));
#pragma unroll
            for (int tm = 0; tm < 2 * QB; ++tm) {
                bf16x8 afrag;
                if (KCH > 0 && tm < 2) afrag = afr[KCH > 0 ? tm * 4 * KCH + kc * 4 + ks : 0];      // kc is a compile-time constant on this path
                else afrag = *reinterpret_cast<const bf16x8*>(smem + ((tm & 1) * 32 + l31) * q_stride + (kc * KSTAGE + ks * 16 + lh * 8) * 2);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[tm], 0, 0, 0);
            }
        }
    };
    // store form with a tile summary: tmax[query][tile] = the largest of the tile's 32 scores (what topk_tiles_rescore_kernel selects on).
    // my_q = the query (of a 64-query block) whose maximum this lane holds after the reduction.
    const int my_q = tile_row_max_query(lane);
    auto tile_done = [&](long t) {
        const long n = t * ROWS_T + l31;
        if (!FILTER && QB == 1 && tmax) {                            // (rows past the gallery's end were loaded as its last row: the maximum is unchanged;
                                                                     //  QB = 2 has no registers left for the reduction: the launcher refuses it)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float v0 = tile_row_max(acc[2 * qb], acc[2 * qb + 1], lane);
                const int qi = qb * 64 + my_q;
                if (qi < B) tmax[(long)qi * ldt + t] = v0;
            }
        }
        if (FILTER) {
#pragma unroll
            for (int tm = 0; tm < 2 * QB; ++tm) {
                if (QB == 2) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(thr_f + tm * 32 + 8 * g4 + 4 * lh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bound[0][4 * g4 + e] = b4[e];
                    }
                }
                filter_tile(acc[tm], bound[QB == 2 ? 0 : tm], tm, n, n < N);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][r] = 0.0f;
            }
        } else {                                                 // sample pass: 128-byte coalesced score stores
            const long grow = sample_row(n < S ? n : S - 1, R);
#pragma unroll
            for (int tm = 0; tm < 2 * QB; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (qi < B && n < S) scores[(long)qi * ld + n] = grow < N ? acc[tm][r] : -__builtin_inff();
                    acc[tm][r] = 0.0f;
                }
        }
    };
    long t_read = gw;
    if (KCH > 0) {
        for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
            for (int kc = 0; kc < (KCH > 0 ? KCH : 1); ++kc) stage_step(kc);
            tile_done(t_read);
            t_read += GW;
        }
    } else {
        int kc = 0;
        for (int s = 0; s < nstages; ++s) {
            stage_step(kc);
            if (++kc == kchunks) { kc = 0; tile_done(t_read); t_read += GW; }
        }
    }
    if (FILTER) flush();
}

hipError_t launch_f32_to_bf16(const float* x, unsigned short* y, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, y, n / 4);
    return hipGetLastError();
}

hipError_t launch_gallery_prepare(const float* x, unsigned short* y, long n, int d, float* meta, hipStream_t s) {
    if (d <= 0 || (d & 3)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(meta, 0, 4 * sizeof(float), s);
    if (e != hipSuccess || n <= 0) return e;
    hipLaunchKernelGGL(gallery_prepare_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, x, y, n, d, meta);
    return hipGetLastError();
}

hipError_t launch_bf16_to_f32(const unsigned short* x, float* y, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, y, n / 4);
    return hipGetLastError();
}

template <int STAGES, bool FILTER, int KCH, int QB = 1>
static hipError_t launch_sweep_inst(const float* q, const unsigned short* g, float* scores, long ld, int B, long N, int D, long S, int R,
                                    const TopkFilter& filt, const int* gate, hipStream_t s, int* zf, float* tmax, long ldt) {
    const size_t qbytes = ((size_t)64 * (D * 2 + 16) + 1023) / 1024 * 1024;
    const size_t ringq = (size_t)4 * STAGES * STAGE_BYTES + QUEUE_BYTES;
    const size_t lds = (KCH > 0 && QB == 1) ? std::max(qbytes, ringq) : qbytes + ringq;
    static size_t attr_set = 0;
    auto kern = sweep_bf16_kernel<STAGES, FILTER, KCH, QB>;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = lds;
    }
    const long ntiles = ((FILTER ? N : S) + ROWS_T - 1) / ROWS_T;
    long blocks = (ntiles + 3) / 4;
    if (blocks > 256) blocks = 256;                                // one persistent workgroup per CU
    FERN_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds, s, q, g, scores, ld, B, N, D, S, R, filt, gate, zf, tmax, ldt);
    return hipGetLastError();
}

template <bool FILTER>
static hipError_t launch_sweep_mode(const float* q, const unsigned short* g, float* scores, long ld, int B, long N, int D, long S, int R,
                                    const TopkFilter& filt, const int* gate, hipStream_t s, int* zf, float* tmax, long ldt) {
    static const bool regq = [] { const char* e = getenv("FERN_SWEEP_REGQ"); return !(e && e[0] == '0'); }();      // A/B switch
    if (B > 64) {    // 65..128 queries per gallery pass: second query block in LDS, 5-stage ring
        switch (D) {
            case 64: return launch_sweep_inst<5, FILTER, 1, 2>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 128: return launch_sweep_inst<5, FILTER, 2, 2>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 256: return launch_sweep_inst<5, FILTER, 4, 2>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 512: return launch_sweep_inst<5, FILTER, 8, 2>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            default: return hipErrorInvalidValue;
        }
    }
    if (regq) {      // register-resident queries, 9-stage ring
        switch (D) {
            case 64: return launch_sweep_inst<9, FILTER, 1>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 128: return launch_sweep_inst<9, FILTER, 2>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 256: return launch_sweep_inst<9, FILTER, 4>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 512: return launch_sweep_inst<9, FILTER, 8>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
            case 640: return launch_sweep_inst<9, FILTER, 10>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);      // RN50x4 (C3): 320 fragment VGPRs of the 512
            default: break;
        }
    }
    const size_t qbytes = ((size_t)64 * (D * 2 + 16) + 1023) / 1024 * 1024;
    const size_t room = (size_t)160 * 1024 - qbytes - QUEUE_BYTES;
    const int stages = (int)(room / (4 * STAGE_BYTES));
    if (stages >= 5) return launch_sweep_inst<5, FILTER, 0>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
    if (stages >= 4) return launch_sweep_inst<4, FILTER, 0>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
    if (stages >= 3) return launch_sweep_inst<3, FILTER, 0>(q, g, scores, ld, B, N, D, S, R, filt, gate, s, zf, tmax, ldt);
    return hipErrorInvalidValue;
}

hipError_t launch_sweep_bf16(const float* q, const unsigned short* g, float* scores, long ld, int B, long N, int D, long S, int R,
                             const TopkFilter* filt, const int* gate, hipStream_t s, int* zero_flags, float* tmax, long ldt) {
    if (B <= 0 || N <= 0) return hipSuccess;
    if (B > 128 || (B > 64 && D != 64 && D != 128 && D != 256 && D != 512) || D % 64 || D > 1024 || R < 1) return hipErrorInvalidValue;
    if (filt) return launch_sweep_mode<true>(q, g, nullptr, 0, B, N, D, 0, 1, *filt, gate, s, nullptr, nullptr, 0);
    if (S <= 0) return hipSuccess;
    if (!scores || (S - 1) * (long)R >= N) return hipErrorInvalidValue;      // every sample run must start inside the gallery
    if (tmax && (S != N || R != 1 || B > 64 || ldt < (N + ROWS_T - 1) / ROWS_T)) return hipErrorInvalidValue;      // tile maxima: store-all form, one query block
    return launch_sweep_mode<false>(q, g, scores, ld, B, N, D, S, R, TopkFilter{}, gate, s, zero_flags, tmax, ldt);
}

}  // namespace fern
