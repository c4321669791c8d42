// Wavefront-reduced exact top-K (K <= 64) over score rows; replaces the reference's full
// torch.argsort of the [Q, N] distance matrix (run/test/test_fiq.py:50), of which only ranks
// < 50 (51 for CIRR) are ever consumed.
//
// Ordering = score descending, gallery index ascending.  Both are packed into one 64-bit key
//     key = orderable(score) << 32 | (0xFFFFFFFF - index)
// so a plain unsigned compare implements the whole rule and ties are deterministic.
//
// A wave keeps its current best-64 list SORTED, ONE ENTRY PER LANE.  Scores are streamed 64 at a time
// (coalesced); a candidate enters only if it beats the list's K-th entry, so after warm-up almost every
// 64-wide step is a single compare + ballot.  An insertion is O(1) wave operations:
// position = popcount(ballot(list > cand)), shift the tail down one lane, drop the candidate in.
// Level 1: grid (segments, B), 4 waves per workgroup each streaming a slice; the 4 lists are merged
// through LDS by streaming them into wave 0's list.  Level 2: one wave per query merges the segment lists.
#include "kernels.h"

namespace fern {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned orderable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float score, unsigned idx) {
    return ((u64)orderable(score) << 32) | (u64)(0xFFFFFFFFu - idx);
}
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

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* scores, long ld, long n, int K, const int* exclude_idx,
                                                        long idx_offset, u64* keys_ws, int nseg, long seg_size) {
    __shared__ u64 lists[4][64];
    const int seg = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* row = scores + (long)b * ld;
    const long seg_lo = (long)seg * seg_size;
    const long seg_hi = seg_lo + seg_size < n ? seg_lo + seg_size : n;
    const long per_wave = seg_size / 4;
    const long lo = seg_lo + wave * per_wave;
    const long hi = lo + per_wave < seg_hi ? lo + per_wave : seg_hi;
    const long excl = exclude_idx ? (long)exclude_idx[b] - idx_offset : -1;   // local column to drop

    u64 best = 0;
    long base = lo;
    for (; base + 256 <= hi; base += 256) {                  // 4 independent coalesced loads in flight, then 4 offers
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = row[base + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long j = base + u * 64 + lane;
            wave_offer(best, j != excl ? make_key(v[u], (unsigned)j) : 0, K, lane);
        }
    }
    for (; base < hi; base += 64) {
        const long j = base + lane;
        u64 cand = 0;
        if (j < hi && j != excl) cand = make_key(row[j], (unsigned)j);
        wave_offer(best, cand, K, lane);
    }
    lists[wave][lane] = best;
    __syncthreads();
    if (wave == 0) {
#pragma unroll 1
        for (int w = 1; w < 4; ++w) best = merge_sorted_desc(best, lists[w][lane], lane);     // the other waves' lists are sorted
        keys_ws[((long)b * nseg + seg) * 64 + lane] = best;
    }
}

// one wave per query: merge `nlists` sorted 64-entry lists, decode to (score, global index)
__global__ __launch_bounds__(64) void topk_final_kernel(const u64* keys_ws, int nlists, int K, long idx_offset, float* out_scores,
                                                        int* out_idx) {
    const int b = blockIdx.x, lane = threadIdx.x;
    u64 best = keys_ws[(long)b * nlists * 64 + lane];
#pragma unroll 1
    for (int l = 1; l < nlists; ++l) best = merge_sorted_desc(best, keys_ws[((long)b * nlists + l) * 64 + lane], lane);
    if (lane < K) {
        float s = -INFINITY;
        int idx = -1;
        if (best != 0) {
            s = unorderable((unsigned)(best >> 32));
            idx = (int)((long)(0xFFFFFFFFu - (unsigned)best) + idx_offset);
        }
        out_scores[(long)b * K + lane] = s;
        out_idx[(long)b * K + lane] = idx;
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

// Scores per level-1 workgroup (4 waves): long enough that the number of per-wave lists -- and with it the total number
// of insertions, ~K (1 + ln(len / K)) per list -- stays small on big galleries, short enough to fill the chip on small ones.
long topk_segment_size(int B, long n) {
    long per_wave = ((long)B * n / 4096 + 63) / 64 * 64;
    per_wave = per_wave < 2048 ? 2048 : (per_wave > 65536 ? 65536 : per_wave);
    return per_wave * 4;
}
int topk_num_segments(int B, long n) {
    const long seg = topk_segment_size(B, n);
    const long k = (n + seg - 1) / seg;
    return (int)(k > 0 ? k : 1);
}

hipError_t launch_topk_rows(const float* scores, long ld, int B, long n, int K, long idx_offset, const int* exclude_idx, u64* keys_ws,
                            float* out_scores, int* out_idx, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || n < 0 || n > 0x7FFFFFF0L) return hipErrorInvalidValue;
    const int nseg = topk_num_segments(B, n);
    hipLaunchKernelGGL(topk_rows_kernel, dim3(nseg, B), dim3(256), 0, s, scores, ld, n, K, exclude_idx, idx_offset, keys_ws, nseg,
                       topk_segment_size(B, n));
    hipLaunchKernelGGL(topk_final_kernel, dim3(B), dim3(64), 0, s, keys_ws, nseg, K, idx_offset, out_scores, out_idx);
    return hipGetLastError();
}

hipError_t launch_topk_merge(const float* scores, const int* idx, float* out_scores, int* out_idx, int R, int B, int K, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (K < 1 || K > 64 || R < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(64), 0, s, scores, idx, out_scores, out_idx, R, B, K);
    return hipGetLastError();
}

}  // namespace fern
