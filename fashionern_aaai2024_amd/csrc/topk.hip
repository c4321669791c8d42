// Wavefront-reduced exact top-K (K <= 64); replaces the reference's full torch.argsort of the [Q, N] distance matrix
// (run/test/test_fiq.py:50), of which only ranks < 50 (51 for CIRR) are ever consumed.
//
// Ordering = score descending, gallery index ascending.  Both are packed into one 64-bit key
//     key = orderable(score) << 32 | (0xFFFFFFFF - index)
// so a plain unsigned compare implements the whole rule and ties are deterministic.
//
// A wave keeps its current best-64 list SORTED, ONE ENTRY PER LANE.  Candidates are streamed 64 at a time (coalesced); a
// candidate enters only if it beats the list's K-th entry, so after warm-up almost every 64-wide step is a single compare +
// ballot.  An insertion is O(1) wave operations: position = popcount(ballot(list > cand)), shift the tail down one lane, drop
// the candidate in; dense batches and whole lists are merged with a bitonic network instead.
// The score matrix is never materialised: the sweep (gemm.hip EPI_TOPK_FILTER / sweep_bf16.hip) appends only the scores that
// reach a per-query bound to candidate lists; the two kernels here compute that bound from a row sample of the gallery
// (topk_sample_bound_kernel) and select the exact top-K from the lists (topk_candidates_kernel).  kernels.h: TopkFilter.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace fern {

typedef unsigned long long u64;

// Cross-lane moves without the LDS crossbar: a wave-uniform source lane is a v_readlane_b32 (SGPR lane select), the
// shift-by-one of the sorted list is a DPP wave_shr:1 move.  ds_bpermute-based __shfl made every insertion a chain of
// ~4 dependent ~120-cycle LDS round trips; these are plain VALU/SALU latencies.
__device__ __forceinline__ u64 shfl64(u64 v, int src) {       // src must be wave-uniform
    const int l = __builtin_amdgcn_readfirstlane(src);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl64v(u64 v, int src) {      // per-lane source
    const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up64(u64 v) {             // lane i <- lane i-1 (lane 0 keeps its value)
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((u64)hi << 32) | lo;
}
// v holds a BITONIC sequence across the lanes: sort it descending (lane 0 = largest) with log2(64) compare-exchange stages.
__device__ __forceinline__ u64 bitonic_finish_desc(u64 v, int lane) {
#pragma unroll
    for (int st = 32; st >= 1; st >>= 1) {
        const u64 o = shfl_xor64(v, st);
        const bool low = (lane & st) == 0;                         // the lower lane of a pair keeps the larger key
        v = low ? (v > o ? v : o) : (v > o ? o : v);
    }
    return v;
}
// Top-64 of the union of two descending-sorted 64-entry lists: max(a[i], b[63-i]) is bitonic and holds exactly those.
__device__ __forceinline__ u64 merge_sorted_desc(u64 a, u64 b_sorted, int lane) {
    const u64 r = shfl64v(b_sorted, 63 - lane);
    return bitonic_finish_desc(a > r ? a : r, lane);
}
// Full bitonic sort (descending) of one key per lane.
__device__ __forceinline__ u64 sort64_desc(u64 v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const u64 o = shfl_xor64(v, j);
            const bool desc = (lane & k) == 0 || k == 64;          // final pass: whole wave descending
            const bool low = (lane & j) == 0;
            const bool keep_max = desc ? low : !low;
            v = keep_max ? (v > o ? v : o) : (v > o ? o : v);
        }
    }
    return v;
}

// Offer one candidate per lane (key 0 = no candidate) to the wave's sorted list `best` (lane i = i-th best).  Few
// survivors of the threshold test are inserted one by one (O(1) wave ops each); many (list still filling) are sorted and
// merged as a batch, which costs the same whatever their number.
__device__ __forceinline__ void wave_offer(u64& best, u64 cand, int K, int lane) {
    u64 thr = shfl64(best, K - 1);
    u64 mask = __ballot(cand > thr);
    if (mask == 0) return;                                        // the common case after warm-up: nothing beats the K-th entry
    if (__popcll(mask) > 6) {
        best = merge_sorted_desc(best, sort64_desc(cand, lane), lane);
        return;
    }
    while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        const u64 c = shfl64(cand, src);
        mask &= mask - 1;
        if (c > thr) {                                            // wave-uniform
            const int pos = __popcll(__ballot(best > c));        // entries that stay ahead of c
            const u64 up = shfl_up64(best);
            best = lane < pos ? best : (lane == pos ? c : up);
            thr = shfl64(best, K - 1);
        }
    }
}

// merge R per-shard lists given as (score, global idx) pairs [R,B,K]
__global__ __launch_bounds__(64) void topk_merge_kernel(const float* scores, const int* idx, float* out_scores, int* out_idx, int R,
                                                        int B, int K) {
    const int b = blockIdx.x, lane = threadIdx.x;
    u64 best = 0;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        u64 cand = 0;
        if (lane < K) {
            const long o = ((long)r * B + b) * K + lane;
            const int gi = idx[o];
            if (gi >= 0) cand = make_key(scores[o], (unsigned)gi);
        }
        wave_offer(best, cand, K, lane);
    }
    if (lane < K) {
        float s = -INFINITY;
        int gi = -1;
        if (best != 0) {
            s = unorderable((unsigned)(best >> 32));
            gi = (int)(0xFFFFFFFFu - (unsigned)best);
        }
        out_scores[(long)b * K + lane] = s;
        out_idx[(long)b * K + lane] = gi;
    }
}

// ---- fused sweep + selection (kernels.h: TopkFilter) -----------------------------------------------------------------
// Both kernels here need "the k-th largest of n keys" for n in the thousands, once per query.  Streaming the keys through the
// sorted wave lists cost 25-55 us (every early candidate is an insertion), a radix select with LDS histograms was no better
// (cosine scores share their top bytes: thousands of LDS atomics on a handful of bins serialise).  What is used instead is
// BISECTION BY COUNTING on keys held in registers: each of the 256 threads keeps <= PER keys, a step counts the keys >= a probe
// (PER compares per thread, a DPP wave sum, four partial sums through LDS, one barrier) and halves the range, which starts at
// [min, max] of the keys -- so the step count is log2 of the keys' spread, not of the key width.  Keys are 64-bit = hi
// (orderable score) << 32 | lo (~index): first the hi word, then lo among the keys that share the k-th hi (usually one key).
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);     // row_mirror: every lane holds its 16-lane row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_inclusive_sum(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d);
        v += lane >= d ? t : 0;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned o = __shfl_xor(v, m); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
// Workgroup helper (NW waves): `red` is 4 * NW words of LDS; `step` alternates the half used so that one barrier per
// reduction is enough (a wave can only reach step + 2 after every wave has read step's sums).
template <int NW>
struct WgReduce {
    int* red;
    int step;
    __device__ __forceinline__ int sum(int local) {
        const int s = wave_sum_i32(local);
        int* slot = red + (step & 1) * NW;
        if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = s;
        __syncthreads();
        ++step;
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += slot[w];
        return t;
    }
    __device__ __forceinline__ unsigned min(unsigned local) {
        const unsigned s = wave_min_u32(local);
        unsigned* slot = reinterpret_cast<unsigned*>(red) + 2 * NW + (step & 1) * NW;
        if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = s;
        __syncthreads();
        ++step;
        unsigned t = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < NW; ++w) t = slot[w] < t ? slot[w] : t;
        return t;
    }
    __device__ __forceinline__ unsigned max(unsigned local) { return ~min(~local); }
};
// Fast path of the selection below.  A bisection step is mostly its barrier (LDS write, barrier, LDS read: ~350 ns with 4 waves,
// more with 16), and 30-odd steps made the bound and select kernels 14-23 us each.  Instead: every wave sorts its 64 lane maxima
// (registers only) and publishes the ceil(k / NW)-th largest; the minimum L0 of those NW values has at least k keys at or above it
// (ceil(k / NW) lanes of every wave hold one), so the k-th largest key lies among the keys with hi >= L0 -- typically a few more
// than k of the thousands.  Those are compacted into LDS (one prefix sum) and ranked by counting (every candidate reads the
// compact list once, broadcast reads): the candidate with k - 1 keys above it is the answer.  Four barriers in all.  More than
// CAPF candidates (tie-heavy galleries: thousands of keys share the k-th score) fall back to the bisection.
__device__ __forceinline__ unsigned sort64_desc_u32(unsigned v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const unsigned o = __shfl_xor(v, j);
            const bool desc = (lane & k) == 0 || k == 64;
            const bool low = (lane & j) == 0;
            const bool keep_max = desc ? low : !low;
            v = keep_max ? (v > o ? v : o) : (v > o ? o : v);
        }
    }
    return v;
}
constexpr int SELECT_CAPF = 512;
// returns true and the k-th largest key in `out` when the fast path applies; every thread must call it (barriers inside);
// ck: SELECT_CAPF keys of LDS, red: 4 * NW + 8 ints of LDS.  The caller has checked that at least k keys exist.
template <int PER, int NW, class LoOf>
__device__ __forceinline__ bool select_kth_fast(const unsigned (&h)[PER], LoOf lo_of, int k, u64* ck, int* red, u64& out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned tmax = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) tmax = h[j] > tmax ? h[j] : tmax;
    const unsigned sorted = sort64_desc_u32(tmax, lane);
    const int tw = (k + NW - 1) / NW;                                   // <= 64: k <= 64
    const unsigned vw = (unsigned)__builtin_amdgcn_readlane((int)sorted, tw - 1);
    unsigned* ured = reinterpret_cast<unsigned*>(red);
    if (lane == 0) ured[wave] = vw;
    __syncthreads();
    unsigned l0 = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < NW; ++w) l0 = ured[w] < l0 ? ured[w] : l0;
    l0 = l0 == 0 ? 1u : l0;                                             // 0 marks an empty slot, never a candidate
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) cnt += h[j] >= l0;
    const int incl = wave_inclusive_sum(cnt, lane);
    if (lane == 63) red[NW + wave] = incl;
    __syncthreads();
    int base = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        base += w < wave ? red[NW + w] : 0;
        total += red[NW + w];
    }
    if (total > SELECT_CAPF) return false;                               // uniform
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (h[j] >= l0) ck[base++] = ((u64)h[j] << 32) | lo_of(j);
    if (tid == 0) { red[2 * NW] = 0; red[2 * NW + 1] = 0; }
    __syncthreads();
    for (int i = tid; i < total; i += NW * 64) {
        const u64 key = ck[i];
        int rank = 0;
        for (int j = 0; j < total; ++j) rank += ck[j] > key;
        if (rank == k - 1) { red[2 * NW] = (int)(unsigned)key; red[2 * NW + 1] = (int)(unsigned)(key >> 32); }
    }
    __syncthreads();
    out = ((u64)(unsigned)red[2 * NW + 1] << 32) | (unsigned)red[2 * NW];
    __syncthreads();                                                     // red / ck may be reused by the caller
    return true;
}

// k-th largest (k >= 1) of the workgroup's keys: thread-local hi words h[0..PER) (0 = empty slot) and lo words from lo_of(j).
// Returns 0 when fewer than k keys exist.  Every thread of the workgroup (NW waves) must call it (barriers inside).  `ck`:
// SELECT_CAPF keys of LDS for the fast path (null: bisection only); red: 4 * NW + 8 ints.
// HAVE_K: the caller knows that at least k keys exist (skips the count and its barrier).
template <int PER, int NW, bool HAVE_K = false, class LoOf>
__device__ __forceinline__ u64 select_kth_largest(const unsigned (&h)[PER], LoOf lo_of, int k, int* red, u64* ck = nullptr) {
    WgReduce<NW> wg{red, 0};
    if (!HAVE_K) {
        int nv = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) nv += h[j] != 0;
        if (wg.sum(nv) < k) return 0;
    }
    if (ck) {
        if (!HAVE_K) __syncthreads();                                    // the reduction above is done with `red`
        u64 fast;
        if (select_kth_fast<PER, NW>(h, lo_of, k, ck, red, fast)) return fast;
        wg.step = 0;
    }
    unsigned mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        mn = (h[j] != 0 && h[j] < mn) ? h[j] : mn;
        mx = h[j] > mx ? h[j] : mx;
    }
    unsigned lo = wg.min(mn), hi = wg.max(mx);
    while (lo < hi) {                                     // largest v with count(h >= v) >= k
        const unsigned mid = lo + (unsigned)(((u64)hi - lo + 1) >> 1);
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) c += h[j] >= mid;
        if (wg.sum(c) >= k) lo = mid;
        else hi = mid - 1;
    }
    const unsigned kh = lo;
    // among the keys with hi == kh, the r-th largest lo (r = k - #keys with a larger hi)
    int gt = 0;
    unsigned lmn = 0xFFFFFFFFu, lmx = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        gt += h[j] > kh;
        if (h[j] == kh) {
            const unsigned l = lo_of(j);
            lmn = l < lmn ? l : lmn;
            lmx = l > lmx ? l : lmx;
        }
    }
    const int r = k - wg.sum(gt);
    unsigned llo = wg.min(lmn), lhi = wg.max(lmx);
    while (llo < lhi) {
        const unsigned mid = llo + (unsigned)(((u64)lhi - llo + 1) >> 1);
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) c += (h[j] == kh) && lo_of(j) >= mid;
        if (wg.sum(c) >= r) llo = mid;
        else lhi = mid - 1;
    }
    return ((u64)kh << 32) | llo;
}

// Step 2: one workgroup per query keeps that query's SAMPLE scores (S columns; column c is gallery row sample_row(c, R)) in
// registers -- 1024 threads, thread t holds columns t, t + 1024, ... (four waves per SIMD keep the compare stream dense) -- and publishes the K-th best key as the query's bound.  Sample columns
// are in gallery order, so a key built from the column index ranks ties like one built from the row; the published key carries
// the real row.  Fewer than K sample rows -> bound 0 (accept all).  Also resets the query's list counters.
// NT threads: 256 (four waves: a barrier among them is a fraction of the 16-wave one, and a bisection step is mostly barrier) for
// samples of <= 4096 rows, 1024 for the larger ones (registers: PER keys per thread).
// Pre-filter form (margin.q != null; api.hip: fern_sim_topk_prefiltered): the sample scores are the bf16 sweep's APPROXIMATIONS s~ of
// the exact fp32 scores s.  With q~ = bf16(q), g~ = bf16(g):  s - s~ = q.(g - g~) + (q - q~).g~  (+ the two accumulations' rounding), so
//     |s - s~| <= eps_b = ||q_b|| E + ||q_b - q~_b|| G~ + slack,   E = max_n ||g_n - g~_n||, G~ = max_n ||g~_n||   (Cauchy-Schwarz)
// for EVERY gallery row -- E, G~, G = max ||g_n|| come from fern_gallery_prepare, the query's two norms are computed here.  If T~ is
// the K-th best approximate score of the whole gallery, every row of the exact top-K has s~ >= T~ - 2 eps (K rows have s >= T~ - eps,
// so the exact K-th best is >= T~ - eps, and a row at or above it has s~ >= s - eps).  The sample's K-th best is <= T~, so the bound
// published for the sweep is (sample K-th best) - margin, margin = 2 eps_b, as a key with the lowest index part; margin_out[b] keeps
// the margin for the rescoring kernel.  slack covers fp32 accumulation in both dot products (D 2^-21 ||q|| max(G, G~): four times
// the textbook D u bound each, the MFMA's internal adder tree is not specified) and the rounding of the norms themselves.
struct BoundMargin {
    const float* q;          // [B, D] fp32 queries (null: no margin -- the sample scores ARE the ranking scores)
    int D;
    const float* meta;       // fern_gallery_prepare: {E, G~, G}
    float* margin_out;       // [B]
};
template <int NT>
__device__ __forceinline__ float bound_margin_of(const BoundMargin& m, int b, float* fred) {
    const int tid = threadIdx.x;
    float a = 0.f, e = 0.f;
    for (int i = tid; i < m.D; i += NT) {
        const float v = m.q[(long)b * m.D + i];
        const float d = v - bf16_bits_to_f32(f32_to_bf16_bits(v));      // exact
        a += v * v;
        e += d * d;
    }
#pragma unroll
    for (int x = 32; x >= 1; x >>= 1) { a += __shfl_xor(a, x); e += __shfl_xor(e, x); }
    if ((tid & 63) == 0) { fred[tid >> 6] = a; fred[NT / 64 + (tid >> 6)] = e; }
    __syncthreads();
    a = 0.f; e = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { a += fred[w]; e += fred[NT / 64 + w]; }
    __syncthreads();
    const float nq = sqrtf(a), eq = sqrtf(e);
    const float E = m.meta[0], Gt = m.meta[1], G = m.meta[2];
    const float eps = (nq * E + eq * Gt) * 1.00390625f + (float)m.D * 4.76837158203125e-7f * nq * fmaxf(G, Gt);      // (1 + 2^-8); D 2^-21
    return 2.0f * eps * 1.0009765625f;      // NaN (a NaN / inf row or query) makes every compare below accept: the exact pass sorts it out
}

template <int PER, int NT>
__global__ __launch_bounds__(NT) void topk_sample_bound_kernel(const float* scores, long ld, long S, int R, int K, const int* exclude,
                                                                long exclude_off, u64* thr_key, int* count, int* flags, int* state, BoundMargin mg) {
    __shared__ int red[80];
    __shared__ u64 ck[SELECT_CAPF];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = scores + (long)b * ld;
    float margin = 0.f;
    if (mg.q) {
        margin = bound_margin_of<NT>(mg, b, reinterpret_cast<float*>(red));
        if (tid == 0) mg.margin_out[b] = margin;
    }
    // sample column of the excluded gallery row (if it was sampled at all): it must not count towards the K rows of the bound
    long drop = -1;
    if (exclude) {
        const long er = (long)exclude[b] - exclude_off;
        if (er >= 0 && er / R < S && sample_row(er / R, R) == er) drop = er / R;
    }
    unsigned h[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const long c = tid + (long)NT * j;
        float v = -INFINITY;
        if (c < S && c != drop) v = row[c];
        h[j] = v == -INFINITY ? 0u : orderable(v);                          // -inf marks padding
    }
    for (int i = tid; i < RANK_SLOTS; i += NT) count[(long)b * RANK_SLOTS + i] = 0;
    const u64 kth = select_kth_largest<PER, NT / 64>(h, [&](int j) { return 0xFFFFFFFFu - (unsigned)(tid + NT * j); }, K, red, ck);
    if (tid == 0) {
        u64 out = 0;
        if (kth != 0) {
            const long c = (long)(0xFFFFFFFFu - (unsigned)kth);
            out = (kth & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - (unsigned)sample_row(c, R));
            if (mg.q) {      // certified pre-filter: lowered by the margin, index part = lowest (every row with that score passes)
                const float lowered = unorderable((unsigned)(kth >> 32)) - margin;
                out = lowered == lowered ? (u64)orderable(lowered) << 32 : 0ull;      // NaN margin: accept all
            }
        }
        thr_key[b] = out;
        state[b] = 0;                      // 0: ranked by the select kernel; 1: lists overflowed -> exact pass; also its done-counter
        state[gridDim.x + b] = 0;
        if (b == 0) { flags[0] = 0; flags[1] = 0; }
    }
}

// Final step: one workgroup per query.  Thread t owns list t; a wave's 64 lists are read with one wave-wide load each and their
// keys dealt round-robin over the 256 threads' registers (LDS as the exchange), the K-th best key comes from the counting
// bisection, the K keys that reach it are collected and one bitonic sort of <= 64 keys writes the ranking.  More than CAND_MAX
// candidates (lists near full) take the streaming path.
constexpr int CAND_PER = 24;                    // keys per thread
constexpr int CAND_MAX = CAND_PER * 256;        // 6144 keys x 8 bytes of LDS
__global__ __launch_bounds__(256) void topk_candidates_kernel(TopkFilter f, int K, long idx_offset, float* out_scores, int* out_idx,
                                                              int* flags, int* state) {
    __shared__ u64 c_key[CAND_MAX];
    __shared__ int red[32];
    __shared__ u64 lists[4][64];
    __shared__ int wtotal[4], over[4], nsel;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt_raw = f.count[(long)b * RANK_SLOTS + tid];      // thread t owns list t
    const int cnt = cnt_raw < f.cap ? cnt_raw : f.cap;
    const bool overflow_w = __any(cnt_raw > f.cap);
    const int incl = wave_inclusive_sum(cnt, lane);
    if (lane == 63) { wtotal[wave] = incl; over[wave] = overflow_w ? 1 : 0; }
    if (tid == 0) nsel = 0;
    __syncthreads();
    if ((over[0] | over[1] | over[2] | over[3]) != 0) {
        // More rows reached the sampled bound than a list holds (a sample that missed a cluster of good rows, a gallery of
        // near-ties): this query goes to the exact pass (rank_exact_kernel), which has no capacity anywhere.
        if (tid == 0) { state[b] = 1; flags[0] = 1; }
        return;
    }
    int base = incl - cnt;
    for (int w = 0; w < wave; ++w) base += wtotal[w];
    const int total = wtotal[0] + wtotal[1] + wtotal[2] + wtotal[3];
    const u64* cand = f.cand + (long)b * RANK_SLOTS * f.cap;
    u64 best = 0;                                        // wave 0 ends up with the sorted top-64 (lane i = i-th best)
    if (total <= CAND_MAX) {
        // gather: every thread walks ITS list (thread t = list t), eight independent loads in flight per round -- lists hold
        // ~K * R / 256 <= 16 keys, so this is one or two round trips -- and drops the keys at its prefix-sum offset in LDS
        int maxc = cnt;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(maxc, m); maxc = o > maxc ? o : maxc; }
        const u64* mylist = cand + (long)tid * f.cap;
#pragma unroll 1
        for (int e0 = 0; e0 < maxc; e0 += 8) {
            u64 key[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) key[u] = e0 + u < cnt ? mylist[e0 + u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u < cnt) c_key[base + e0 + u] = key[u];
        }
        __syncthreads();
        u64 mine[CAND_PER];
        unsigned h[CAND_PER];
#pragma unroll
        for (int j = 0; j < CAND_PER; ++j) {
            const int i = tid + 256 * j;
            mine[j] = i < total ? c_key[i] : 0;
            h[j] = (unsigned)(mine[j] >> 32);
        }
        const int want = total < K ? total : K;          // fewer candidates than K: all of them rank
        u64 kth = 0;
        __syncthreads();                                 // every thread holds its keys in registers: c_key's head becomes the fast path's compact list
        if (total > 64) kth = select_kth_largest<CAND_PER, 4>(h, [&](int j) { return (unsigned)mine[j]; }, want, red, c_key);
#pragma unroll
        for (int j = 0; j < CAND_PER; ++j) {
            if (mine[j] != 0 && mine[j] >= kth) {
                const int p = atomicAdd(&nsel, 1);
                if (p < 64) lists[0][p] = mine[j];
            }
        }
        __syncthreads();
        if (wave != 0) return;
        const int ns = nsel < 64 ? nsel : 64;
        best = sort64_desc(lane < ns ? lists[0][lane] : 0, lane);
    } else {
        // streaming path (lists near full: > CAND_MAX candidates): the sorted wave lists of wave_offer
#pragma unroll 1
        for (int j = 0; j < 64; ++j) {
            const int n = __builtin_amdgcn_readlane(cnt, j);
            wave_offer(best, lane < n ? cand[(long)(wave * 64 + j) * f.cap + lane] : 0, K, lane);
        }
        lists[wave][lane] = best;
        __syncthreads();
        if (wave != 0) return;
#pragma unroll 1
        for (int w = 1; w < 4; ++w) best = merge_sorted_desc(best, lists[w][lane], lane);
    }
    if (lane < K) {
        float sc = -INFINITY;
        int idx = -1;
        if (best != 0) {
            sc = unorderable((unsigned)(best >> 32));
            idx = (int)((long)(0xFFFFFFFFu - (unsigned)best) + idx_offset);
        }
        out_scores[(long)b * K + lane] = sc;
        out_idx[(long)b * K + lane] = idx;
    }
}

typedef float f32x4e __attribute__((ext_vector_type(4)));
// ---- certified pre-filter: select on the approximate keys, then rescore the survivors exactly ---------------------------------------
// Final step of fern_sim_topk_prefiltered.  The candidate lists hold APPROXIMATE keys (bf16 sweep scores s~) of every row with
// s~ >= (sample bound) - margin.  Per query (one workgroup): gather the lists as topk_candidates_kernel does, T~ = the K-th best
// approximate score (exact, of the whole gallery: every row at or above the sample bound is in the lists), keep the rows with
// s~ >= T~ - margin -- a superset of the exact top-K (topk_sample_bound_kernel's comment has the argument) -- and give each survivor
// its EXACT fp32 score: one sequential v_fma_f32 chain per (query, row) in the k order of the fp32 MFMA kernels (8g, 8g+4, 8g+1, 8g+5,
// ...: oracle/chain.c), i.e. bit for bit the score fern_sim_topk's sweep produces.  The exact keys are then ranked.  Output = the
// exact top-K with the exact scores; the approximate scores decide nothing but which ~K + (rows within the margin) rows get rescored.
// Rescoring loads are COALESCED: a wave takes 64 survivors at a time and walks D in 64-float chunks; 16 lanes fetch one row's 256-byte
// chunk (4 rows per load instruction), the chunk tile [64 rows][64 k] is transposed through the wave's LDS tile and lane l then runs
// survivor l's chain over its 64 k.  (One lane loading its own row touches 64 cache lines per instruction: 8x the address work.)
// Anything without room -- a list overflow, more than CAND_MAX candidates, more than RESC_MAX survivors (galleries of near-ties) --
// goes to the exact pass (rank_exact_kernel on the fp32 gallery), whose bound (sample bound - margin) is a valid lower bound of the
// exact K-th best too.
constexpr int RESC_MAX = 1024;                  // survivors per query that are rescored here
constexpr int RESC_PER = RESC_MAX / 256;
constexpr int RESC_CH = 32;                     // k per step of the transposing tile
constexpr int RESC_TLD = RESC_CH + 4;           // floats per tile row (+ 4: lane l's ds_read_b128 of row l starts 4 banks after lane l-1's)
constexpr int RESC_ROWS = 32;                   // survivors per wave and round
constexpr int RESC_TILE_FLOATS = 4 * 2 * RESC_ROWS * RESC_TLD;      // the four waves' tiles, double buffered

// One wave, 32 survivors surv[s0 .. s0 + 32) (rows past ns redo survivor s0): their exact fp32 chains against qrow.  D = 32 nsteps k,
// nsteps % RING == 0.  A step is four load instructions (8 rows x 32 floats each: whole 128-byte lines) whose registers go through
// the wave's LDS tile so that lane l < 32 can continue survivor l's chain over those 32 k.  The loads of the next RING steps are always
// in flight -- a register ring, a slot refilled right after its registers were written to the tile -- and the tile is double buffered:
// step c + 1 is written while step c is read.  The loop body is BRANCH-FREE (loads past the row's end re-read its last step, the
// tile write after the last step is never read): behind `if (c < nsteps)` guards the compiler waited for every load right after
// issuing it (s_waitcnt vmcnt(0..3) in front of each tile write) and the ring was one deep.
template <int RING>
__device__ __forceinline__ float rescore_rows(const unsigned* surv, int s0, int ns, const float* qrow, const float* gallery, int D, float* tl) {
    const int lane = threadIdx.x & 63;
    const int lrow = lane >> 3, lcol = (lane & 7) * 4;       // loader role: row inside a group of 8, float offset inside a 32-float step
    const int l31 = lane & 31;
    const int nsteps = D / RESC_CH;
    const float* src[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int r = s0 + rg * 8 + lrow;
        src[rg] = gallery + (long)surv[r < ns ? r : s0] * D + lcol;
    }
    f32x4e reg[RING][4];
#pragma unroll
    for (int c = 0; c < RING; ++c)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) reg[c][rg] = *reinterpret_cast<const f32x4e*>(src[rg] + c * RESC_CH);      // nsteps >= RING
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) *reinterpret_cast<f32x4e*>(tl + (rg * 8 + lrow) * RESC_TLD + lcol) = reg[0][rg];      // step 0's tile
    {
        const int cn = RING < nsteps ? RING : nsteps - 1;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) reg[0][rg] = *reinterpret_cast<const f32x4e*>(src[rg] + cn * RESC_CH);
    }
    float acc = 0.0f;
    for (int c0 = 0; c0 < nsteps; c0 += RING) {
#pragma unroll
        for (int cc = 0; cc < RING; ++cc) {
            const int c = c0 + cc;
            constexpr int TS = RESC_ROWS * RESC_TLD;
            const int nslot = (cc + 1) % RING;
            float* nt = tl + ((cc + 1) & 1) * TS;            // RING is even: step c + 1's buffer parity is static
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) *reinterpret_cast<f32x4e*>(nt + (rg * 8 + lrow) * RESC_TLD + lcol) = reg[nslot][rg];
            {
                const int cn = c + 1 + RING < nsteps ? c + 1 + RING : nsteps - 1;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) reg[nslot][rg] = *reinterpret_cast<const f32x4e*>(src[rg] + cn * RESC_CH);
            }
            const float* mrow = tl + (cc & 1) * TS + l31 * RESC_TLD;
            const float* qk = qrow + c * RESC_CH;
            f32x4e gg[RESC_CH / 4], qq[RESC_CH / 4];
#pragma unroll
            for (int g4 = 0; g4 < RESC_CH / 4; ++g4) {
                gg[g4] = *reinterpret_cast<const f32x4e*>(mrow + g4 * 4);
                qq[g4] = *reinterpret_cast<const f32x4e*>(qk + g4 * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this step's reads (and the next step's tile writes) are done
#pragma unroll
            for (int g8 = 0; g8 < RESC_CH / 8; ++g8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_fmaf(qq[2 * g8][e], gg[2 * g8][e], acc);
                    acc = __builtin_fmaf(qq[2 * g8 + 1][e], gg[2 * g8 + 1][e], acc);
                }
            }
        }
    }
    return acc;
}

// Exact scores of the survivors surv[0..ns) of query row `qrow` (LDS) and their ranking: the tail shared by the two rescoring
// kernels.  Every thread of the 256-thread workgroup calls it.  A wave takes 32 survivors per round (rescore_rows).
// Ranking: every thread counts the keys above its survivor's key (LDS broadcast reads) and writes rank < K straight to its slot -- no
// second selection, no sort.  D % 64 == 0.
__device__ __forceinline__ void rescore_and_rank(const unsigned* surv, int ns, const float* qrow, const float* gallery, int D, float* tiles,
                                                 u64* x_key, int K, long idx_offset, float* out_scores, int* out_idx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* tl = tiles + wave * (2 * RESC_ROWS * RESC_TLD);
    const int nsteps = D / RESC_CH;
    for (int s0 = wave * RESC_ROWS; s0 < ns; s0 += 4 * RESC_ROWS) {
        float acc;
        if (nsteps % 8 == 0) acc = rescore_rows<8>(surv, s0, ns, qrow, gallery, D, tl);
        else if (nsteps % 10 == 0) acc = rescore_rows<10>(surv, s0, ns, qrow, gallery, D, tl);      // D = 640 (RN50x4): 20 steps, 40 loads in flight
        else if (nsteps % 4 == 0) acc = rescore_rows<4>(surv, s0, ns, qrow, gallery, D, tl);
        else acc = rescore_rows<2>(surv, s0, ns, qrow, gallery, D, tl);
        if (lane < RESC_ROWS && s0 + lane < ns) x_key[s0 + lane] = make_key(acc, surv[s0 + lane]);
    }
    if (tid < 4) x_key[ns + tid] = 0;                // the counting loop reads four keys at a time: zeros never count (x_key holds RESC_MAX + 4)
    __syncthreads();
    // rank by counting: keys are distinct (the index is part of the key); four keys per pair of LDS reads
    for (int i = tid; i < ns; i += 256) {
        const u64 key = x_key[i];
        int rank = 0;
        for (int j0 = 0; j0 < ns; j0 += 4) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(x_key + j0), c = *reinterpret_cast<const ulonglong2*>(x_key + j0 + 2);
            rank += (a.x > key) + (a.y > key) + (c.x > key) + (c.y > key);
        }
        if (rank < K) {
            out_scores[rank] = unorderable((unsigned)(key >> 32));
            out_idx[rank] = (int)((long)(0xFFFFFFFFu - (unsigned)key) + idx_offset);
        }
    }
    for (int r = ns + tid; r < K; r += 256) { out_scores[r] = -INFINITY; out_idx[r] = -1; }      // fewer survivors than K: the gallery ran out
}

static_assert(RESC_TILE_FLOATS * 4 <= CAND_MAX * 8, "the four waves' (double-buffered) tiles overlay the candidate keys");
__global__ __launch_bounds__(256) void topk_rescore_kernel(TopkFilter f, const float* q, const float* gallery, int D, const float* margin_in, int K,
                                                           long idx_offset, float* out_scores, int* out_idx, int* flags, int* state) {
    __shared__ __attribute__((aligned(16))) u64 c_key[CAND_MAX];      // candidates (approximate keys); then select_kth_fast's compact list; then the waves' tiles
    __shared__ __attribute__((aligned(16))) u64 x_key[RESC_MAX + 4];      // exact keys of the survivors
    __shared__ unsigned surv[RESC_MAX];         // gallery rows of the survivors
    __shared__ __attribute__((aligned(16))) float qrow[1024];
    __shared__ int red[32];
    __shared__ int wtotal[4], over[4], nsurv;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64* mylist = f.cand + ((long)b * RANK_SLOTS + tid) * f.cap;      // thread t owns list t
    // the list's first entries are requested together with its count (entries past the count are stale and dropped below): one
    // round trip instead of two for the ~K R / 256 entries a list typically holds
    constexpr int SPEC = 8;
    u64 first[SPEC];
#pragma unroll
    for (int u = 0; u < SPEC; ++u) first[u] = u < f.cap ? mylist[u] : 0;
    const int cnt_raw = f.count[(long)b * RANK_SLOTS + tid];
    const int cnt = cnt_raw < f.cap ? cnt_raw : f.cap;
    const bool overflow_w = __any(cnt_raw > f.cap);
    const int incl = wave_inclusive_sum(cnt, lane);
    if (lane == 63) { wtotal[wave] = incl; over[wave] = overflow_w ? 1 : 0; }
    if (tid == 0) nsurv = 0;
    for (int i = tid; i < D; i += 256) qrow[i] = q[(long)b * D + i];
    __syncthreads();
    const int total = wtotal[0] + wtotal[1] + wtotal[2] + wtotal[3];
    if ((over[0] | over[1] | over[2] | over[3]) != 0 || total > CAND_MAX) {
        if (tid == 0) { state[b] = 1; flags[0] = 1; }
        return;
    }
    int base = incl - cnt;
    for (int w = 0; w < wave; ++w) base += wtotal[w];
    {
#pragma unroll
        for (int u = 0; u < SPEC; ++u)
            if (u < cnt) c_key[base + u] = first[u];
        int maxc = cnt;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(maxc, m); maxc = o > maxc ? o : maxc; }
#pragma unroll 1
        for (int e0 = SPEC; e0 < maxc; e0 += 8) {
            u64 key[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) key[u] = e0 + u < cnt ? mylist[e0 + u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u < cnt) c_key[base + e0 + u] = key[u];
        }
    }
    __syncthreads();
    u64 mine[CAND_PER];
    unsigned h[CAND_PER];
#pragma unroll
    for (int j = 0; j < CAND_PER; ++j) {
        const int i = tid + 256 * j;
        mine[j] = i < total ? c_key[i] : 0;
        h[j] = (unsigned)(mine[j] >> 32);
    }
    const int want = total < K ? total : K;
    u64 kth = 0;
    __syncthreads();                                 // keys are in registers: c_key's head becomes the fast path's compact list
    if (total > K) kth = select_kth_largest<CAND_PER, 4, true>(h, [&](int j) { return (unsigned)mine[j]; }, want, red, c_key);
    // survivors: approximate score >= T~ - margin (total <= K: every candidate); a NaN on either side keeps the row
    const float cut = kth != 0 ? unorderable((unsigned)(kth >> 32)) - margin_in[b] : -INFINITY;
#pragma unroll
    for (int j = 0; j < CAND_PER; ++j) {
        if (mine[j] != 0 && !(unorderable(h[j]) < cut)) {
            const int p = atomicAdd(&nsurv, 1);
            if (p < RESC_MAX) surv[p] = 0xFFFFFFFFu - (unsigned)mine[j];
        }
    }
    __syncthreads();                                 // also: nobody reads c_key as keys any more -- it becomes the tiles
    const int ns = nsurv;
    if (ns > RESC_MAX) {
        if (tid == 0) { state[b] = 1; flags[0] = 1; }
        return;
    }
    rescore_and_rank(surv, ns, qrow, gallery, D, reinterpret_cast<float*>(c_key), x_key, K, idx_offset, out_scores + (long)b * K, out_idx + (long)b * K);
}

// ---- dense form of the certified pre-filter (galleries up to DENSE_MAX_N rows) -------------------------------------------------------
// For a small gallery the stage is launch boundaries, not bytes (C2: five kernels, 76 us, of which the sweep's 47 MB are ~10).  Here
// the bf16 sweep simply STORES its approximate scores ([B, ld] fp32: a quarter of the bf16 gallery's bytes at B = 64, D = 512) -- no
// sample pass, no bound kernel, no candidate lists, no atomics -- and this kernel does everything else, one workgroup per query:
//   pass 1  every thread's maximum over its strided share of the row -> per wave the ceil(K / 4)-th largest lane maximum -> l0 = the
//           smallest of the four: at least K scores are >= l0, so T~ (the K-th best approximate score) >= l0;
//   pass 2  (the row again, from L2) collect the rows with s~ >= l0 - margin: a superset of {s~ >= T~ - margin};
//   T~ among the collected, survivors = collected rows with s~ >= T~ - margin, exact rescoring + ranking (rescore_and_rank).
// Without room (more than DENSE_CAP collected, more than RESC_MAX survivors) the query is flagged for the exact pass with the bound
// it had reached (l0 - margin or T~ - margin: lower bounds of the exact K-th best).  Resets done[b] for that pass; flags[0] is
// zeroed by the host launcher before the sweep.
// Exact top-K of ONE query by its own workgroup (256 threads), for the dense form on SMALL galleries: a query the pre-filter has no room
// for (near-tie floods, galleries smaller than ~K, NaN scores) is ranked right here -- every row's exact chain, streamed through the
// waves' sorted lists (wave_offer: no capacity) -- instead of by the gated exact-pass launch, whose ~4.5 us of launch boundary every call
// paid for a case that almost never happens.  One CU streams the whole fp32 gallery (~60 GB/s): the launcher allows it up to 128 MB.
__device__ __forceinline__ void exact_topk_inline(const float* qrow, const float* gallery, long N, int D, long drop, int K, long idx_offset,
                                                  float* os, int* oi, u64 (*wl)[64]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u64 best = 0;
#pragma unroll 1
    for (long n0 = (long)wave * 64; n0 < N; n0 += 256) {
        const long n = n0 + lane;
        u64 key = 0;
        if (n < N && n != drop) {
            const float* g = gallery + n * D;
            float acc = 0.0f;
            for (int k8 = 0; k8 < D; k8 += 8) {
                const f32x4e g0 = *reinterpret_cast<const f32x4e*>(g + k8), g1 = *reinterpret_cast<const f32x4e*>(g + k8 + 4);
                const f32x4e q0 = *reinterpret_cast<const f32x4e*>(qrow + k8), q1 = *reinterpret_cast<const f32x4e*>(qrow + k8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_fmaf(q0[e], g0[e], acc);
                    acc = __builtin_fmaf(q1[e], g1[e], acc);
                }
            }
            key = make_key(acc, (unsigned)n);
        }
        wave_offer(best, key, K, lane);
    }
    wl[wave][lane] = best;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (int w = 1; w < 4; ++w) best = merge_sorted_desc(best, wl[w][lane], lane);
    if (lane < K) {
        float sc = -INFINITY;
        int idx = -1;
        if (best != 0) {
            sc = unorderable((unsigned)(best >> 32));
            idx = (int)((long)(0xFFFFFFFFu - (unsigned)best) + idx_offset);
        }
        os[lane] = sc;
        oi[lane] = idx;
    }
}

constexpr int DENSE_SEG = 1024;                 // collected keys per wave (4 segments)
constexpr int DENSE_UB = 16;                    // 16-byte loads a thread keeps in flight while it walks the score row
constexpr int DENSE_BATCH = 1024 * DENSE_UB;    // floats the workgroup covers per batch
// NB > 0: the row fits NB batches, which stay in registers between the two passes (N <= NB * 16384); NB = 0: any N, the row is read
// twice (the second time from this XCD's L2).
template <int NB>
__global__ __launch_bounds__(256) void topk_dense_rescore_kernel(const float* approx, long ld, long N, const float* q, const float* gallery, int D,
                                                                 BoundMargin mg, int K, const int* exclude, long exclude_off, long idx_offset,
                                                                 float* out_scores, int* out_idx, u64* thr_key, int* flags, int* state, int* done, int stop, int inline_exact) {
    __shared__ u64 wlists[4][64];                                            // inline exact ranking: the waves' sorted lists
    __shared__ __attribute__((aligned(16))) u64 ckey[4][DENSE_SEG + 2];      // per wave: collected approximate keys (+ zero padding)
    __shared__ __attribute__((aligned(16))) float tiles[RESC_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) u64 x_key[RESC_MAX + 4];
    __shared__ unsigned surv[RESC_MAX];
    __shared__ __attribute__((aligned(16))) float qrow[1024];
    __shared__ int red[32];
    __shared__ unsigned wmax[4];
    __shared__ int wcnt[4];
    __shared__ int nsurv;
    __shared__ unsigned kth_hi;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = approx + (long)b * ld;
    if (tid == 0) { nsurv = 0; kth_hi = 0; state[b] = 0; done[b] = 0; }
    for (int i = tid; i < D; i += 256) qrow[i] = q[(long)b * D + i];
    long drop = -1;
    if (exclude) {
        const long er = (long)exclude[b] - exclude_off;
        if (er >= 0 && er < N) drop = er;
    }
    const long n4 = N & ~3L;
    constexpr int NBR = NB > 0 ? NB : 1;
    f32x4e v[NBR][DENSE_UB];
    // This kernel runs ONE wave per SIMD (64 workgroups on 64 CUs): every VALU instruction of a phase is 4 cycles of its latency, so
    // the walks over the row are written for instruction count -- float maxima (v_max3_f32) instead of orderable keys, the excluded
    // row handled by a separate (rare) path, keys built only for the ~100 collected rows.
    // One batch: DENSE_UB loads per thread, UNCONDITIONAL on a clamped address and masked afterwards (behind `if (i < n4)` the
    // compiler put every load in its own branch with an s_waitcnt vmcnt(0) in front: sixteen serial round trips per batch).  The
    // row was written by the sweep from other XCDs: it comes from HBM / the Infinity Cache at 1-2 us a round trip.
    // Loaded values are SANITISED once (positions past the row and the excluded row become -inf), so that the two walks are one
    // v_max3 / one v_cmp per element.
    auto load_batch = [&](long base, f32x4e (&dst)[DENSE_UB]) {
#pragma unroll
        for (int u = 0; u < DENSE_UB; ++u) {
            const long i = base + u * 1024L + tid * 4;
            dst[u] = *reinterpret_cast<const f32x4e*>(row + (i < n4 ? i : 0));
        }
#pragma unroll
        for (int u = 0; u < DENSE_UB; ++u) {
            const long i = base + u * 1024L + tid * 4;
            const bool in = i < n4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[u][e] = in ? dst[u][e] : -INFINITY;
            if (drop >= i && drop < i + 4) dst[u][drop - i] = -INFINITY;     // (rare: the excluded row)
        }
    };
    // the query's two norms (bound_margin_of): loads issued now, reduced after pass 1
    float qa = 0.f, qe = 0.f;
    for (int i = tid; i < D; i += 256) {
        const float x = q[(long)b * D + i];
        const float dd = x - bf16_bits_to_f32(f32_to_bf16_bits(x));
        qa += x * x;
        qe += dd * dd;
    }
    // pass 1: every thread's maximum score.  fmax drops NaNs; the running SUM of the scores is NaN whenever one of them is (or when
    // +inf meets -inf: a false alarm that only costs speed) -- a NaN anywhere makes the wave collect everything.
    float fmax_t = -INFINITY, nsum = 0.f;
    auto max_batch = [&](const f32x4e (&src)[DENSE_UB]) {
#pragma unroll
        for (int u = 0; u < DENSE_UB; ++u) {
            fmax_t = fmaxf(fmaxf(fmax_t, src[u][0]), fmaxf(src[u][1], fmaxf(src[u][2], src[u][3])));
            nsum += (src[u][0] + src[u][1]) + (src[u][2] + src[u][3]);      // (sanitised -inf positions keep the sum at -inf, not NaN)
        }
    };
    if (NB > 0) {
#pragma unroll
        for (int nb = 0; nb < NBR; ++nb) load_batch((long)nb * DENSE_BATCH, v[nb]);
#pragma unroll
        for (int nb = 0; nb < NBR; ++nb) max_batch(v[nb]);
    } else {
        for (long base = 0; base < n4; base += DENSE_BATCH) {
            load_batch(base, v[0]);
            max_batch(v[0]);
        }
    }
    float tail = -INFINITY;                          // the row's last N % 4 scores, one per thread (-inf: none / excluded)
    if (tid < (int)(N - n4) && n4 + tid != drop) { tail = row[n4 + tid]; fmax_t = fmaxf(fmax_t, tail); nsum += tail; }
    const bool nan_t = nsum != nsum;
    unsigned tmax = fmax_t == -INFINITY ? 0u : orderable(fmax_t);      // 0 = "no row": a thread without rows, or only -inf scores
    if (stop == 1) { if (tmax == 0x12345) out_idx[0] = 1; return; }
    // margin (bound_margin_of's arithmetic on the sums started above)
    float margin;
    {
        float* fred = reinterpret_cast<float*>(red);
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) { qa += __shfl_xor(qa, x); qe += __shfl_xor(qe, x); }
        if (lane == 0) { fred[wave] = qa; fred[4 + wave] = qe; }
        __syncthreads();
        const float a = fred[0] + fred[1] + fred[2] + fred[3], e = fred[4] + fred[5] + fred[6] + fred[7];
        const float nq = sqrtf(a), eq = sqrtf(e);
        const float E = mg.meta[0], Gt = mg.meta[1], G = mg.meta[2];
        const float eps = (nq * E + eq * Gt) * 1.00390625f + (float)D * 4.76837158203125e-7f * nq * fmaxf(G, Gt);
        margin = 2.0f * eps * 1.0009765625f;
    }
    const unsigned sorted = sort64_desc_u32(tmax, lane);
    const int tw = (K + 3) / 4;
    const unsigned vw = (unsigned)__builtin_amdgcn_readlane((int)sorted, tw - 1);
    if (lane == 0) wmax[wave] = __any(nan_t) ? 0u : vw;                 // a NaN score anywhere in the wave's share: collect everything
    __syncthreads();
    unsigned l0 = wmax[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) l0 = wmax[w] < l0 ? wmax[w] : l0;
    // l0 == 0: some wave holds fewer than ceil(K / 4) rows (a gallery of < ~K rows), or met a NaN score
    // A query the pre-filter has no room for leaves the do-block with `fallback_thr` = the bound it had reached (a lower bound of the
    // exact K-th best) and is ranked exactly behind it: ONE copy of that code, at the end.
    u64 fallback_thr = 0;
    do {
    if (l0 == 0) break;      // (uniform) no bound -- tiny galleries, or NaN scores
    const float cut0 = unorderable(l0) - margin;
    if (stop == 2) { if (cut0 == 0.12345f) out_idx[0] = 1; return; }
    // (This kernel runs ONCE per CU on a cold instruction cache: its time follows the bytes of code it executes.  A second, never-taken
    // copy of the collection code (an "accept everything" form for tiny galleries) cost 4 us; those queries go to the exact pass instead.)
    // pass 2: collect the rows at or above cut0 into THIS WAVE's segment -- one compare + ballot per element, no atomics (one returning
    // LDS atomic per hit was ~200 cycles each); stored as raw (score bits, row), keyed after the pass
    u64* seg = ckey[wave];
    int wn = 0;                                      // wave-uniform: entries in the segment so far
    auto collect_one = [&](float val, long n, bool hit) {
        const u64 m = __ballot(hit);
        if (m) {
            const int pos = wn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (hit && pos < DENSE_SEG) seg[pos] = ((u64)__float_as_uint(val) << 32) | (unsigned)n;
            wn += __popcll(m);
        }
    };
    auto collect_batch = [&](long base, const f32x4e (&src)[DENSE_UB]) {
#pragma unroll
        for (int u = 0; u < DENSE_UB; ++u) {
            const long i = base + u * 1024L + tid * 4;
            {
                // two v_max3 + one compare reject a position (64 lanes x 4 scores) that holds nothing at or above the cut -- about half of them
                const float m4 = fmaxf(fmaxf(src[u][0], src[u][1]), fmaxf(src[u][2], src[u][3]));
                const bool nan4 = (src[u][0] + src[u][1]) + (src[u][2] + src[u][3]) != (src[u][0] + src[u][1]) + (src[u][2] + src[u][3]);
                if (__ballot(!(m4 < cut0) || nan4) == 0) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) collect_one(src[u][e], i + e, !(src[u][e] < cut0));      // sanitised positions are -inf < cut0
            }
        }
    };
    if (stop == 21) {
    } else if (NB > 0) {
#pragma unroll
        for (int nb = 0; nb < NBR; ++nb) collect_batch((long)nb * DENSE_BATCH, v[nb]);
    } else {
        for (long base = 0; base < n4; base += DENSE_BATCH) {
            load_batch(base, v[0]);
            collect_batch(base, v[0]);
        }
    }
    collect_one(tail, n4 + tid, tid < (int)(N - n4) && n4 + tid != drop && !(tail < cut0));
    // raw entries -> ranking keys (orderable score, ~row), in place: the wave's own entries, no barrier needed before
    for (int i = lane; i < (wn < DENSE_SEG ? wn : DENSE_SEG); i += 64) {
        const u64 r = seg[i];
        seg[i] = make_key(__uint_as_float((unsigned)(r >> 32)), (unsigned)r);
    }
    const int wkeep = wn < DENSE_SEG ? wn : DENSE_SEG;
    if (lane < 2) seg[wkeep + lane] = 0;             // zero padding: the counting loop reads two keys at a time
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    const int c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
    const int nc = c0 + c1 + c2 + c3;
    if (stop == 3 || stop == 21) { if (tid == 0) out_idx[(long)b * K] = nc; return; }
    if (c0 > DENSE_SEG || c1 > DENSE_SEG || c2 > DENSE_SEG || c3 > DENSE_SEG) {
        fallback_thr = cut0 == cut0 ? (u64)orderable(cut0) << 32 : 0ull;
        break;
    }
    // the i-th collected key (segments in wave order)
    auto key_at = [&](int i) -> u64 {
        if (i < c0) return ckey[0][i];
        i -= c0;
        if (i < c1) return ckey[1][i];
        i -= c1;
        if (i < c2) return ckey[2][i];
        return ckey[3][i - c2];
    };
    // T~ = K-th best approximate key of the collected rows (= of the gallery), by counting: keys are distinct, one 64-bit compare per
    // pair (the scores and rows as separate words cost ~15 VALU instructions per pair: 10 us at 127 collected rows)
    unsigned khi = 0;
    if (nc > K) {
        for (int i = tid; i < nc; i += 256) {
            const u64 key = key_at(i);
            int rank = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int cw = w == 0 ? c0 : w == 1 ? c1 : w == 2 ? c2 : c3;
                for (int j0 = 0; j0 < cw; j0 += 2) {
                    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&ckey[w][j0]);
                    rank += (a.x > key) + (a.y > key);
                }
            }
            if (rank == K - 1) kth_hi = (unsigned)(key >> 32);
        }
        if (stop == 36) return;
        __syncthreads();
        khi = kth_hi;
    }
    const float cut = khi != 0 ? unorderable(khi) - margin : -INFINITY;
    if (stop == 4) { if (cut == 0.12345f) out_idx[0] = 1; return; }
    for (int i = tid; i < nc; i += 256) {
        const u64 key = key_at(i);
        if (!(unorderable((unsigned)(key >> 32)) < cut)) {
            const int p = atomicAdd(&nsurv, 1);
            if (p < RESC_MAX) surv[p] = 0xFFFFFFFFu - (unsigned)key;
        }
    }
    __syncthreads();
    const int ns = nsurv;
    if (stop == 5) { if (tid == 0) out_idx[(long)b * K] = ns; return; }
    if (ns > RESC_MAX) {
        fallback_thr = cut == cut && khi != 0 ? (u64)orderable(cut) << 32 : 0ull;
        break;
    }
    rescore_and_rank(surv, ns, qrow, gallery, D, tiles, x_key, K, idx_offset, out_scores + (long)b * K, out_idx + (long)b * K);
    return;
    } while (0);
    if (inline_exact) {      // small galleries: ranked exactly right here
        exact_topk_inline(qrow, gallery, N, D, drop, K, idx_offset, out_scores + (long)b * K, out_idx + (long)b * K, wlists);
    } else if (tid == 0) {   // flagged for the gated exact pass
        state[b] = 1;
        flags[0] = 1;
        thr_key[b] = fallback_thr;
    }
}

// ---- dense form on tile maxima (galleries past TILES_MIN_N rows) -------------------------------------------------------------------
// topk_dense_rescore_kernel walks a query's whole score row twice with ONE workgroup: 2 x 184 KB at C2 (13 us of its 24), 2 x 800 KB
// at C3 (~60 of ~75).  Here the sweep also leaves tmax[b][t] = the largest score of gallery tile t (32 consecutive rows: one wave tile
// of sweep_bf16_kernel, reduced across its lanes), and the kernel reads N / 32 numbers instead of N:
//   bound   every thread's maximum over its strided share of the TILE maxima -> per wave the ceil(K' / 4)-th largest -> l0 = the
//           smallest of the four: K' distinct tiles hold a score >= l0 (K' = K + 1 when a row is excluded: it may be one of them),
//           so T~ (the K-th best approximate score) >= l0;
//   tiles   the tiles with tmax >= l0 - margin (~one per collected row: ~130 of 1 438 at C2) are listed, per wave, by ballot;
//   gather  only those tiles' scores are read (128 bytes each, all loads of the workgroup in flight at once) and the rows at or above
//           l0 - margin collected -- the same set the row walk collected;
//   then T~, survivors, exact rescoring and ranking exactly as topk_dense_rescore_kernel.
// A margin that is not finite (NaN / inf anywhere in the query or the gallery: launch_gallery_prepare poisons meta) sends the query to
// the exact ranking at once; with a finite margin every score is finite (|s~| <= ||q|| G~ (1 + D 2^-24)), so fmax loses nothing.
constexpr int TILES_TB = 32;                    // tile maxima a thread holds per batch (8 192 tiles = 262 144 rows per batch)
constexpr int TILES_SEG = 512;                  // listed tiles per wave
constexpr long TILES_MIN_N = 16384;             // below: the row walk (a wave needs ceil(K' / 4) tiles of its own for the bound)
template <bool HOLD>
__global__ __launch_bounds__(256) void topk_tiles_rescore_kernel(const float* approx, long ld, const float* tmax, long ldt, long N, const float* q,
                                                                 const float* gallery, int D, BoundMargin mg, int K, const int* exclude, long exclude_off,
                                                                 long idx_offset, float* out_scores, int* out_idx, u64* thr_key, int* flags, int* state,
                                                                 int* done, int inline_exact) {
    __shared__ u64 wlists[4][64];
    __shared__ __attribute__((aligned(16))) u64 ckey[4][DENSE_SEG + 2];
    __shared__ __attribute__((aligned(16))) float tiles[RESC_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) u64 x_key[RESC_MAX + 4];
    __shared__ unsigned surv[RESC_MAX];
    __shared__ __attribute__((aligned(16))) float qrow[1024];
    __shared__ int tlist[4][TILES_SEG];
    __shared__ int red[32];
    __shared__ unsigned wmax[4];
    __shared__ int wcnt[4], tcnt[4];
    __shared__ int nsurv;
    __shared__ unsigned kth_hi;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = approx + (long)b * ld;
    const float* trow = tmax + (long)b * ldt;
    const long ntiles = (N + 31) >> 5;
    if (tid == 0) { nsurv = 0; kth_hi = 0; state[b] = 0; done[b] = 0; }
    for (int i = tid; i < D; i += 256) qrow[i] = q[(long)b * D + i];
    long drop = -1;
    if (exclude) {
        const long er = (long)exclude[b] - exclude_off;
        if (er >= 0 && er < N) drop = er;
    }
    // tile maxima: TILES_TB unconditional loads on clamped addresses (topk_dense_rescore_kernel's note), masked afterwards
    float tv[TILES_TB];
    auto load_tiles = [&](long base) {
#pragma unroll
        for (int u = 0; u < TILES_TB; ++u) {
            const long t = base + u * 256 + tid;
            tv[u] = trow[t < ntiles ? t : 0];
        }
#pragma unroll
        for (int u = 0; u < TILES_TB; ++u) tv[u] = base + u * 256 + tid < ntiles ? tv[u] : -INFINITY;
    };
    float qa = 0.f, qe = 0.f;
    for (int i = tid; i < D; i += 256) {
        const float x = q[(long)b * D + i];
        const float dd = x - bf16_bits_to_f32(f32_to_bf16_bits(x));
        qa += x * x;
        qe += dd * dd;
    }
    float fmax_t = -INFINITY;
    for (long base = 0; base < (HOLD ? 1 : ntiles); base += 256 * TILES_TB) {
        load_tiles(base);
#pragma unroll
        for (int u = 0; u < TILES_TB; u += 2) fmax_t = fmaxf(fmax_t, fmaxf(tv[u], tv[u + 1]));
    }
    const unsigned tmx = fmax_t == -INFINITY ? 0u : orderable(fmax_t);      // 0 = "no tile"
    float margin;
    {
        float* fred = reinterpret_cast<float*>(red);
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) { qa += __shfl_xor(qa, x); qe += __shfl_xor(qe, x); }
        if (lane == 0) { fred[wave] = qa; fred[4 + wave] = qe; }
        __syncthreads();
        const float a = fred[0] + fred[1] + fred[2] + fred[3], e = fred[4] + fred[5] + fred[6] + fred[7];
        const float nq = sqrtf(a), eq = sqrtf(e);
        const float E = mg.meta[0], Gt = mg.meta[1], G = mg.meta[2];
        const float eps = (nq * E + eq * Gt) * 1.00390625f + (float)D * 4.76837158203125e-7f * nq * fmaxf(G, Gt);
        margin = 2.0f * eps * 1.0009765625f;
    }
    const unsigned sorted = sort64_desc_u32(tmx, lane);
    const int tw = (K + (drop >= 0 ? 1 : 0) + 3) / 4;
    const unsigned vw = (unsigned)__builtin_amdgcn_readlane((int)sorted, tw - 1);
    if (lane == 0) wmax[wave] = vw;
    __syncthreads();
    unsigned l0 = wmax[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) l0 = wmax[w] < l0 ? wmax[w] : l0;
    u64 fallback_thr = 0;
    do {
    if (l0 == 0 || !(margin < INFINITY)) break;      // (uniform) no bound: a wave without ceil(K' / 4) tiles; or no certificate
    const float cut0 = unorderable(l0) - margin;
    // the tiles that can hold a row at or above cut0, per wave by ballot
    int* tl = tlist[wave];
    int tn = 0;                                      // wave-uniform
    for (long base = 0; base < (HOLD ? 1 : ntiles); base += 256 * TILES_TB) {
        if (!HOLD) load_tiles(base);
#pragma unroll
        for (int u = 0; u < TILES_TB; ++u) {
            const bool hit = !(tv[u] < cut0);        // masked positions are -inf < cut0 (cut0 is finite)
            const u64 m = __ballot(hit);
            if (m) {
                const int pos = tn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                if (hit && pos < TILES_SEG) tl[pos] = (int)(base + u * 256 + tid);
                tn += __popcll(m);
            }
        }
    }
    if (lane == 0) tcnt[wave] = tn;
    __syncthreads();
    const int t0 = tcnt[0], t1 = tcnt[1], t2 = tcnt[2], t3 = tcnt[3];
    if (t0 > TILES_SEG || t1 > TILES_SEG || t2 > TILES_SEG || t3 > TILES_SEG) {      // near-tie floods: the exact ranking, behind the bound
        fallback_thr = (u64)orderable(cut0) << 32;
        break;
    }
    const int p1 = t0, p2 = t0 + t1, p3 = p2 + t2, nt = p3 + t3;
    // gather: 8 lanes x 16 bytes per listed tile, four rounds of loads in flight; hits go to the wave's segment as (score bits, row)
    u64* seg = ckey[wave];
    int wn = 0;
    const int items = nt * 8;
    for (int e0 = 0; e0 < items; e0 += 1024) {
        f32x4e val[4];
        long nn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 + tid, ec = e < items ? e : 0, slot = ec >> 3;
            const int w = (slot >= p1) + (slot >= p2) + (slot >= p3);
            const int tile = tlist[w][slot - (w == 0 ? 0 : w == 1 ? p1 : w == 2 ? p2 : p3)];
            nn[u] = (long)tile * 32 + (ec & 7) * 4;
            val[u] = *reinterpret_cast<const f32x4e*>(row + nn[u]);      // ld covers whole tiles (launcher)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = e0 + u * 256 + tid < items;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = nn[u] + j;
                const bool hit = in && n < N && n != drop && !(val[u][j] < cut0);
                const u64 m = __ballot(hit);
                if (m) {
                    const int pos = wn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    if (hit && pos < DENSE_SEG) seg[pos] = make_key(val[u][j], (unsigned)n);
                    wn += __popcll(m);
                }
            }
        }
    }
    const int wkeep = wn < DENSE_SEG ? wn : DENSE_SEG;
    if (lane < 2) seg[wkeep + lane] = 0;
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    const int c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
    const int nc = c0 + c1 + c2 + c3;
    if (c0 > DENSE_SEG || c1 > DENSE_SEG || c2 > DENSE_SEG || c3 > DENSE_SEG) {
        fallback_thr = (u64)orderable(cut0) << 32;
        break;
    }
    auto key_at = [&](int i) -> u64 {
        if (i < c0) return ckey[0][i];
        i -= c0;
        if (i < c1) return ckey[1][i];
        i -= c1;
        if (i < c2) return ckey[2][i];
        return ckey[3][i - c2];
    };
    unsigned khi = 0;
    if (nc > K) {
        for (int i = tid; i < nc; i += 256) {
            const u64 key = key_at(i);
            int rank = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int cw = w == 0 ? c0 : w == 1 ? c1 : w == 2 ? c2 : c3;
                for (int j0 = 0; j0 < cw; j0 += 2) {
                    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&ckey[w][j0]);
                    rank += (a.x > key) + (a.y > key);
                }
            }
            if (rank == K - 1) kth_hi = (unsigned)(key >> 32);
        }
        __syncthreads();
        khi = kth_hi;
    }
    const float cut = khi != 0 ? unorderable(khi) - margin : -INFINITY;
    for (int i = tid; i < nc; i += 256) {
        const u64 key = key_at(i);
        if (!(unorderable((unsigned)(key >> 32)) < cut)) {
            const int p = atomicAdd(&nsurv, 1);
            if (p < RESC_MAX) surv[p] = 0xFFFFFFFFu - (unsigned)key;
        }
    }
    __syncthreads();
    const int ns = nsurv;
    if (ns > RESC_MAX) {
        fallback_thr = khi != 0 ? (u64)orderable(cut) << 32 : 0ull;
        break;
    }
    rescore_and_rank(surv, ns, qrow, gallery, D, tiles, x_key, K, idx_offset, out_scores + (long)b * K, out_idx + (long)b * K);
    return;
    } while (0);
    if (inline_exact) {
        exact_topk_inline(qrow, gallery, N, D, drop, K, idx_offset, out_scores + (long)b * K, out_idx + (long)b * K, wlists);
    } else if (tid == 0) {
        state[b] = 1;
        flags[0] = 1;
        thr_key[b] = fallback_thr;
    }
}

// ---- exact pass ---------------------------------------------------------------------------------------------------------------
// Runs (gated on flags[0]) for the queries the select kernel sent here.  One launch of `groups` workgroups; workgroup g, wave w
// owns the 32-row gallery tiles (4g + w) + j * 4 * groups.  The flagged queries are compacted (in query order) and taken 32 AT A
// TIME as the 32 A rows of the tile MFMAs -- ONE gallery pass per 32 flagged queries (round 3 broadcast one query to all 32 rows:
// a pass per query, 1/32 of the MFMA).  A tile's scores come from the SAME MFMA sequence as the sweep kernels -- fp32:
// v_mfma_f32_32x32x2_f32 per 8-group g8 and e = 0..3 with lane half h feeding k = 8 g8 + 4 h + e (gemm.hip); bf16:
// v_mfma_f32_32x32x16_bf16 per 16-k step with lane half h feeding k = 16 ks + 8 h .. + 7, the query rounded to bf16 like
// sweep_bf16.hip does -- so register r of lane (n & 31) + 32 h holds the bit-identical score of gallery row n for query
// (r & 3) + 8 (r >> 2) + 4 h of the chunk.  Keys that reach a query's (still valid) sampled bound are offered to that query's sorted
// list of the wave (64 entries, one per lane, kept in LDS: lists[wave][query]); after the pass the four wave lists of a query are
// merged into partial[b][g], and the last workgroup to finish a query (a ticket in done[b]) merges the partial lists and writes the
// ranking.  Nothing here has a capacity: whatever the gallery looks like, the result is the exact top-K.
typedef float f32x16e __attribute__((ext_vector_type(16)));
typedef short bf16x8e __attribute__((ext_vector_type(8)));
constexpr int EXACT_QB = 32;                    // flagged queries per gallery pass (the A rows of one MFMA tile)
template <bool BF16>
__global__ __launch_bounds__(256) void rank_exact_kernel(const float* q, const void* gallery, int B, long N, int D, int K, const int* state,
                                                         const u64* thr_key, const int* exclude, long exclude_off, long idx_offset,
                                                         u64* partial, int* done, float* out_scores, int* out_idx, const int* gate) {
    if (*gate == 0) return;
    __shared__ u64 lists[4][EXACT_QB][64];      // 64 KiB: every wave's sorted list of every query of the chunk
    __shared__ int flagged[1024];               // the flagged queries in query order (a plan holds <= 1024 queries: api.hip kRankQueryChunk)
    __shared__ u64 thr_s[EXACT_QB];
    __shared__ long ex_s[EXACT_QB];
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int G = gridDim.x, g = blockIdx.x;
    const long ntiles = (N + 31) / 32;
    // compaction: thread t looks at queries 4t .. 4t + 3
    int fl[4], cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int b = tid * 4 + u;
        fl[u] = b < B && state[b] != 0;
        cnt += fl[u];
    }
    const int incl = wave_inclusive_sum(cnt, lane);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - cnt;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    const int nflag = wtot[0] + wtot[1] + wtot[2] + wtot[3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (fl[u]) flagged[base++] = tid * 4 + u;
    __syncthreads();
    for (int c0 = 0; c0 < nflag; c0 += EXACT_QB) {
        const int nq = nflag - c0 < EXACT_QB ? nflag - c0 : EXACT_QB;       // queries of this chunk (uniform)
        if (tid < EXACT_QB) {
            const int b = flagged[c0 + (tid < nq ? tid : 0)];
            thr_s[tid] = tid < nq ? thr_key[b] : ~0ull;
            ex_s[tid] = (tid < nq && exclude) ? (long)exclude[b] - exclude_off : -1;
        }
        for (int i = tid; i < 4 * EXACT_QB * 64; i += 256) (&lists[0][0][0])[i] = 0;
        __syncthreads();
        const float* qrow = q + (long)flagged[c0 + (l31 < nq ? l31 : 0)] * D;      // this lane's A row
        float bound[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bound[r] = filter_bound(thr_s[(r & 3) + 8 * (r >> 2) + 4 * lh]);      // +inf for the unused rows
        for (long t = (long)g * 4 + wave; t < ntiles; t += (long)G * 4) {
            const long n = t * 32 + l31;
            const long nc = n < N ? n : N - 1;
            f32x16e acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if (BF16) {
                const unsigned short* grow = reinterpret_cast<const unsigned short*>(gallery) + nc * D;
                for (int ks = 0; ks < D / 16; ++ks) {
                    const f32x4e a0 = *reinterpret_cast<const f32x4e*>(qrow + ks * 16 + lh * 8);
                    const f32x4e a1 = *reinterpret_cast<const f32x4e*>(qrow + ks * 16 + lh * 8 + 4);
                    bf16x8e af;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[e] = (short)f32_to_bf16_bits(a0[e]); af[4 + e] = (short)f32_to_bf16_bits(a1[e]); }
                    const bf16x8e bf = *reinterpret_cast<const bf16x8e*>(grow + ks * 16 + lh * 8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                }
            } else {
                const float* grow = reinterpret_cast<const float*>(gallery) + nc * D;
                for (int g8 = 0; g8 < D / 8; ++g8) {
                    const f32x4e af = *reinterpret_cast<const f32x4e*>(qrow + g8 * 8 + lh * 4);
                    const f32x4e bf = *reinterpret_cast<const f32x4e*>(grow + g8 * 8 + lh * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
                }
            }
            // which queries of the chunk have a score at or above their bound in this tile (wave-uniform bit mask)
            unsigned qmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const u64 m = __ballot(n < N && !(acc[r] < bound[r]));
                const int qa = (r & 3) + 8 * (r >> 2);
                qmask |= ((unsigned)m != 0u ? 1u : 0u) << qa;
                qmask |= ((unsigned)(m >> 32) != 0u ? 1u : 0u) << (qa + 4);
            }
            qmask &= nq >= 32 ? 0xFFFFFFFFu : ((1u << nq) - 1u);
            while (qmask) {                                        // wave-uniform loop over the queries with survivors
                const int qi = __builtin_ctz(qmask);
                qmask &= qmask - 1;
                const int rsel = (qi & 3) + 4 * (qi >> 3), hsel = (qi >> 2) & 1;
                float v = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) v = rsel == r ? acc[r] : v;
                u64 cand = 0;
                if (lh == hsel && n < N && n != ex_s[qi]) {
                    const u64 key = make_key(v, (unsigned)n);
                    cand = key >= thr_s[qi] ? key : 0;
                }
                u64 best = lists[wave][qi][lane];
                wave_offer(best, cand, K, lane);
                lists[wave][qi][lane] = best;
            }
        }
        __syncthreads();
        // wave w finishes queries w, w + 4, ...: merge the four wave lists, publish the partial list, take a ticket
        for (int qi = wave; qi < nq; qi += 4) {
            const int b = flagged[c0 + qi];
            u64 best = lists[0][qi][lane];
#pragma unroll 1
            for (int w = 1; w < 4; ++w) best = merge_sorted_desc(best, lists[w][qi][lane], lane);
            partial[((long)b * G + g) * 64 + lane] = best;
            __threadfence();
            int ticket = 0;
            if (lane == 0) ticket = atomicAdd(&done[b], 1);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == G - 1) {                                  // every other workgroup's partial list is visible (its fence precedes its ticket)
                __threadfence();
                u64 top = 0;
#pragma unroll 1
                for (int gg = 0; gg < G; ++gg) {
                    const u64 other = __hip_atomic_load(&partial[((long)b * G + gg) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    top = gg == 0 ? other : merge_sorted_desc(top, other, lane);
                }
                if (lane < K) {
                    float sc = -INFINITY;
                    int idx = -1;
                    if (top != 0) {
                        sc = unorderable((unsigned)(top >> 32));
                        idx = (int)((long)(0xFFFFFFFFu - (unsigned)top) + idx_offset);
                    }
                    out_scores[(long)b * K + lane] = sc;
                    out_idx[(long)b * K + lane] = idx;
                }
            }
        }
        __syncthreads();                                            // the lists are re-zeroed for the next chunk
    }
}

hipError_t launch_topk_sample_bound(const float* scores, long ld, int B, long S, int R, int K, const int* exclude, long exclude_off,
                                    u64* thr_key, int* count, int* flags, int* state, hipStream_t s, const float* q, int D, const float* meta,
                                    float* margin_out) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || S < 0 || R < 1 || (q && (!meta || !margin_out || D <= 0))) return hipErrorInvalidValue;
    const BoundMargin mg{q, D, meta, margin_out};
    // keys per thread of the bisection (registers): the plan caps S at 32768 (api.hip: rank_plan)
    auto go = [&](auto kern, int nt) { FERN_LAUNCH(kern, dim3(B), dim3(nt), 0, s, scores, ld, S, R, K, exclude, exclude_off, thr_key, count, flags, state, mg); };
    if (S <= 1024) go(topk_sample_bound_kernel<4, 256>, 256);
    else if (S <= 4096) go(topk_sample_bound_kernel<16, 256>, 256);
    else if (S <= 16 * 1024) go(topk_sample_bound_kernel<16, 1024>, 1024);
    else if (S <= 32 * 1024) go(topk_sample_bound_kernel<32, 1024>, 1024);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_topk_candidates(const TopkFilter& f, int B, int K, long idx_offset, float* out_scores, int* out_idx, int* flags,
                                  int* state, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || f.cap < 1 || f.cap > 64) return hipErrorInvalidValue;
    FERN_LAUNCH(topk_candidates_kernel, dim3(B), dim3(256), 0, s, f, K, idx_offset, out_scores, out_idx, flags, state);
    return hipGetLastError();
}

hipError_t launch_topk_rescore(const TopkFilter& f, const float* q, const float* gallery, int D, const float* margin, int B, int K, long idx_offset,
                               float* out_scores, int* out_idx, int* flags, int* state, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || f.cap < 1 || f.cap > 64 || D < 64 || D % 64 || D > 1024 || !margin) return hipErrorInvalidValue;
    FERN_LAUNCH(topk_rescore_kernel, dim3(B), dim3(256), 0, s, f, q, gallery, D, margin, K, idx_offset, out_scores, out_idx, flags, state);
    return hipGetLastError();
}

hipError_t launch_topk_dense_rescore(const float* approx, long ld, long N, const float* q, const float* gallery, int D, const float* meta,
                                     int B, int K, const int* exclude, long exclude_off, long idx_offset, float* out_scores,
                                     int* out_idx, unsigned long long* thr_key, int* flags, int* state, int* done, hipStream_t s, int inline_exact,
                                     const float* tmax, long ldt) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || D < 64 || D % 64 || D > 1024 || N < 1 || (ld & 3) || !meta) return hipErrorInvalidValue;
    const BoundMargin mg{q, D, meta, nullptr};
    if (tmax && N >= TILES_MIN_N) {                  // selection on the sweep's tile maxima
        if (ld < ((N + 31) & ~31L) || ldt < (N + 31) / 32) return hipErrorInvalidValue;
        if ((N + 31) / 32 <= 256L * TILES_TB)
            FERN_LAUNCH(topk_tiles_rescore_kernel<true>, dim3(B), dim3(256), 0, s, approx, ld, tmax, ldt, N, q, gallery, D, mg, K, exclude, exclude_off,
                        idx_offset, out_scores, out_idx, thr_key, flags, state, done, inline_exact);
        else
            FERN_LAUNCH(topk_tiles_rescore_kernel<false>, dim3(B), dim3(256), 0, s, approx, ld, tmax, ldt, N, q, gallery, D, mg, K, exclude, exclude_off,
                        idx_offset, out_scores, out_idx, thr_key, flags, state, done, inline_exact);
        return hipGetLastError();
    }
    // lab switch (tools/rank_bench.py): FERN_DENSE_STOP=k ends the kernel after phase k (results are then garbage) to attribute its time
    static const int stop = [] { const char* e = getenv("FERN_DENSE_STOP"); return e ? atoi(e) : 0; }();
    const long n4 = N & ~3L;
#define FERN_DENSE_GO(NB)                                                                                                                       \
    FERN_LAUNCH(topk_dense_rescore_kernel<NB>, dim3(B), dim3(256), 0, s, approx, ld, N, q, gallery, D, mg, K, exclude, exclude_off, idx_offset, \
                out_scores, out_idx, thr_key, flags, state, done, stop, inline_exact)
    if (n4 <= 1L * DENSE_BATCH) FERN_DENSE_GO(1);
    else if (n4 <= 2L * DENSE_BATCH) FERN_DENSE_GO(2);
    else if (n4 <= 3L * DENSE_BATCH) FERN_DENSE_GO(3);
    else FERN_DENSE_GO(0);
#undef FERN_DENSE_GO
    return hipGetLastError();
}

hipError_t launch_rank_exact(const float* q, const void* gallery, int gallery_bf16, int B, long N, int D, int K, const int* state,
                             const unsigned long long* thr_key, const int* exclude, long exclude_off, long idx_offset,
                             unsigned long long* partial, int groups, int* done, float* out_scores, int* out_idx, const int* gate,
                             hipStream_t s) {
    if (B <= 0 || N <= 0) return hipSuccess;
    if (K < 1 || K > 64 || groups < 1 || B > 1024 || D % (gallery_bf16 ? 16 : 8)) return hipErrorInvalidValue;
    if (gallery_bf16)
        FERN_LAUNCH(rank_exact_kernel<true>, dim3(groups), dim3(256), 0, s, q, gallery, B, N, D, K, state, thr_key, exclude, exclude_off,
                    idx_offset, partial, done, out_scores, out_idx, gate);
    else
        FERN_LAUNCH(rank_exact_kernel<false>, dim3(groups), dim3(256), 0, s, q, gallery, B, N, D, K, state, thr_key, exclude, exclude_off,
                    idx_offset, partial, done, out_scores, out_idx, gate);
    return hipGetLastError();
}

hipError_t launch_topk_merge(const float* scores, const int* idx, float* out_scores, int* out_idx, int R, int B, int K, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || R < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(64), 0, s, scores, idx, out_scores, out_idx, R, B, K);
    return hipGetLastError();
}

}  // namespace fern
