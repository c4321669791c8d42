// fp32 MFMA attention for short sequences (<= 224 keys): softmax(scale * Q K^T [+causal]) V.
//
// Replaces nn.MultiheadAttention / HF BertSelfAttention / CLIP attention on the hot path
// (SURVEY.md 2.2 rows K2-K4): ViT-B/16 (197 tokens, 12 heads x 64), CLIP text (77, causal, 8 x 64),
// the fusion BERT block (91 tokens, 8 heads x 64 or 80) and the 13 x 13 cross attention.
//
// One workgroup = one (batch, head).  K and V of the head live in LDS for the whole workgroup
// (every sequence on this path fits), each wave owns 32-query tiles:
//   * keys are walked in 32-key tiles with an online (running max / running sum) softmax, so only one
//     score tile is live in registers at a time;
//   * S^T = K Q^T on v_mfma_f32_32x32x2_f32 with K as the A operand: the accumulator then has the
//     QUERY on the lane and the KEYS in registers, so the softmax row reduction is lane-local plus
//     one cross-half shuffle, and
//   * the un-normalised P^T accumulator registers are, as they stand, the B operand of
//     O^T = V^T P^T (register r of lane half h holds key (r&3) + 8(r>>2) + 4h -- exactly the k pair
//     one 32x32x2 step consumes), so P never leaves registers.
// K rows are padded to HDP+4 floats so the ds_read_b128 fragment reads are bank-conflict free.
#include "kernels.h"

#include <cstdlib>

namespace fern {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HDP, int NT, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn_f32_kernel(AttnParams p) {
    constexpr int KS = HDP + 4;          // K row stride in LDS (floats)
    constexpr int ROWS = NT * 32;        // padded key count
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                    // [ROWS][KS]
    float* Vs = smem + ROWS * KS;        // [ROWS][HDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int hd = p.hd;

    // ---- stage K and V of this (batch, head) into LDS, zero padded ----
    {
        constexpr int C4 = HDP / 4;
        const float* kb = p.k + (long)b * p.s_k * p.ldk + (long)h * hd;
        const float* vb = p.v + (long)b * p.s_k * p.ldv + (long)h * hd;
        for (int i = tid; i < ROWS * C4; i += NW * 64) {
            const int row = i / C4, c = (i % C4) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (row < p.s_k && c < hd) {
                kv = *reinterpret_cast<const f32x4*>(kb + (long)row * p.ldk + c);
                vv = *reinterpret_cast<const f32x4*>(vb + (long)row * p.ldv + c);
            }
            *reinterpret_cast<f32x4*>(&Ks[row * KS + c]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[row * HDP + c]) = vv;
        }
    }
    __syncthreads();

    const int nqt = (p.s_q + 31) / 32;
    for (int qt = wave; qt < nqt; qt += NW) {
        const int qi = qt * 32 + l31;                 // this lane's query
        const int qrow = qi < p.s_q ? qi : p.s_q - 1;
        // ---- Q fragment (B operand): lane (query, half) holds d = 8*kk + 4*half + e, pre-scaled ----
        f32x4 qf[HDP / 8];
        {
            const float* qb = p.q + ((long)b * p.s_q + qrow) * p.ldq + (long)h * hd;
#pragma unroll
            for (int kk = 0; kk < HDP / 8; ++kk) {
                const int d = kk * 8 + 4 * lh;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (d < hd) t = *reinterpret_cast<const f32x4*>(qb + d);
                qf[kk] = t * p.scale;
            }
        }
        // ---- online softmax over 32-key tiles: only one S^T tile is live at a time ----
        f32x16 o[HDP / 32];
#pragma unroll
        for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
        float m = -INFINITY, sum = 0.0f;          // sum: this lane half's partial; halves share m
        const int nt = CAUSAL ? (qt + 1 < NT ? qt + 1 : NT) : NT;   // causal: later tiles are all in the future
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            // S^T tile: st[r] = score(key 32t + (r&3) + 8(r>>2) + 4*half, query)
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < HDP / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(t * 32 + l31) * KS + kk * 8 + 4 * lh]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[kk][e], st, 0, 0, 0);
            }
            float mt = -INFINITY;
            if (CAUSAL || (t + 1) * 32 > p.s_k) {            // only the last key tile (or a causal one) has keys to mask: wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool ok = key < p.s_k && (!CAUSAL || key <= qi);
                    st[r] = ok ? st[r] : -INFINITY;
                    mt = fmaxf(mt, st[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m, mt);      // finite from tile 0 on: key 0 is valid for every query
            const float alpha = __expf(m - m_new);   // exp(-inf) = 0 on the first tile
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[r] - m_new);   // masked entries: exp(-inf) = 0
                st[r] = e;
                ps += e;
            }
            sum = sum * alpha + ps;
            m = m_new;
            // O^T = alpha * O^T + V_t^T P_t^T: A = V^T[d][key]; B = the P^T registers as they stand
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float vf = Vs[key * HDP + db * 32 + l31];
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, st[r], o[db], 0, 0, 0);
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // ---- store: lane (query, half) holds d = 32*db + 8*g + 4*half + {0..3} in registers 4g..4g+3 ----
        if (qi < p.s_q && p.out_b) {
            unsigned short* ob = p.out_b + ((long)b * p.s_q + qi) * p.ldo + (long)h * hd;
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * lh;
                    if (d < hd) {
                        ushort4 t;
                        t.x = f32_to_bf16_bits(o[db][4 * g] * inv); t.y = f32_to_bf16_bits(o[db][4 * g + 1] * inv);
                        t.z = f32_to_bf16_bits(o[db][4 * g + 2] * inv); t.w = f32_to_bf16_bits(o[db][4 * g + 3] * inv);
                        *reinterpret_cast<ushort4*>(ob + d) = t;
                    }
                }
        } else if (qi < p.s_q) {
            float* ob = p.out + ((long)b * p.s_q + qi) * p.ldo + (long)h * hd;
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * lh;
                    if (d < hd) {
                        f32x4 t = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
                        *reinterpret_cast<f32x4*>(ob + d) = t;
                    }
                }
        }
    }
}

// ---- key-chunked form (197-token ViT heads) -----------------------------------------------------------------------------
// The kernel above keeps ALL keys of a head in LDS: 224 padded rows x (68 + 64) floats = 118 KB, so one workgroup fills a CU
// (160 KB) and its phases -- staging (memory), S^T and P V (MFMA), softmax (VALU) -- run back to back with nothing beside them:
// the phases of the 197 x 197 x 64 shape add up (32 + 42 + 34 + 13 us measured with parts switched off), and while an attention
// workgroup is resident only one 32 KB GEMM workgroup of another stream fits next to it.  Here the keys are staged in KH chunks of
// CT 32-key tiles (4 + 3 tiles for 197 keys: 68 KB), the online softmax simply carries (m, sum, O) across the chunk boundary, and
// TWO workgroups (or one and two GEMM workgroups) share a CU: one's staging overlaps the other's MFMAs, four waves per SIMD cover
// each other's softmax.  Each wave owns at most one 32-query tile (launch condition: s_q <= 32 NW).  Arithmetic per (query, key)
// is the kernel above's, in the same order: bit-identical results.
template <int HDP, int NT, int NW, int KH>
__global__ __launch_bounds__(NW * 64, 2) void attn_f32_chunked_kernel(AttnParams p) {
    constexpr int KS = HDP + 4;
    constexpr int CT = (NT + KH - 1) / KH;      // key tiles per chunk
    constexpr int ROWS = CT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                    // [ROWS][KS]
    float* Vs = smem + ROWS * KS;        // [ROWS][HDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int hd = p.hd;
    const bool active = wave * 32 < p.s_q;
    const int qi = wave * 32 + l31;
    const int qrow = qi < p.s_q ? qi : p.s_q - 1;
    f32x4 qf[HDP / 8];
    {
        const float* qb = p.q + ((long)b * p.s_q + qrow) * p.ldq + (long)h * hd;
#pragma unroll
        for (int kk = 0; kk < HDP / 8; ++kk) {
            const int d = kk * 8 + 4 * lh;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (active && d < hd) t = *reinterpret_cast<const f32x4*>(qb + d);
            qf[kk] = t * p.scale;
        }
    }
    f32x16 o[HDP / 32];
#pragma unroll
    for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m = -INFINITY, sum = 0.0f;
    const float* kb = p.k + (long)b * p.s_k * p.ldk + (long)h * hd;
    const float* vb = p.v + (long)b * p.s_k * p.ldv + (long)h * hd;
#pragma unroll 1
    for (int c = 0; c < KH; ++c) {
        if (c) __syncthreads();                                      // every wave is done with the previous chunk
        {
            constexpr int C4 = HDP / 4;
            for (int i = tid; i < ROWS * C4; i += NW * 64) {
                const int row = i / C4, cc = (i % C4) * 4, key = c * ROWS + row;
                f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
                if (key < p.s_k && cc < hd) {
                    kv = *reinterpret_cast<const f32x4*>(kb + (long)key * p.ldk + cc);
                    vv = *reinterpret_cast<const f32x4*>(vb + (long)key * p.ldv + cc);
                }
                *reinterpret_cast<f32x4*>(&Ks[row * KS + cc]) = kv;
                *reinterpret_cast<f32x4*>(&Vs[row * HDP + cc]) = vv;
            }
        }
        __syncthreads();
        if (!active) continue;
        const int nt = (c + 1) * CT <= NT ? CT : NT - c * CT;
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < HDP / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(t * 32 + l31) * KS + kk * 8 + 4 * lh]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[kk][e], st, 0, 0, 0);
            }
            float mt = -INFINITY;
            if ((c * CT + t + 1) * 32 > p.s_k) {             // only the last key tile has keys to mask: wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (c * CT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    st[r] = key < p.s_k ? st[r] : -INFINITY;
                    mt = fmaxf(mt, st[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m, mt);
            const float alpha = __expf(m - m_new);
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[r] - m_new);
                st[r] = e;
                ps += e;
            }
            sum = sum * alpha + ps;
            m = m_new;
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float vf = Vs[row * HDP + db * 32 + l31];
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, st[r], o[db], 0, 0, 0);
                }
            }
        }
    }
    if (!active || qi >= p.s_q) return;
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    float* ob = p.out + ((long)b * p.s_q + qi) * p.ldo + (long)h * hd;
#pragma unroll
    for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + 8 * g + 4 * lh;
            if (d < hd) {
                f32x4 t = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(ob + d) = t;
            }
        }
}

// ---- bf16 operand form ---------------------------------------------------------------------------------------------
// Same structure on v_mfma_f32_32x32x16_bf16 (perf mode of the CLIP towers): S^T = K Q^T with K as the A operand, so the
// accumulator has the query on the lane and 16 keys in registers; registers 8s..8s+7, rounded to bf16, ARE the B operand
// of k-step s of O^T = V^T P^T -- with the k order permuted: element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3).
// The matching V^T fragment (d on the lane, those 8 keys in the elements) comes from two ds_read_b64_tr_b16 transposing
// reads of the row-major V image.  K image: rows padded to HDP*2+16 bytes (conflict-free ds_read_b128); V image:
// [HDP/32][key][32 columns] with 64-byte rows (the 4 x 64-byte block one half-wave transposes covers all 64 banks once).
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int HDP, int NT, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bf16_kernel(AttnParams p) {
    constexpr int KSB = HDP * 2 + 16;    // K row stride in LDS (bytes)
    constexpr int ROWS = NT * 32;        // padded key count
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char* Ks = smem_b;                   // [ROWS][KSB]
    char* Vs = smem_b + ROWS * KSB;      // [HDP/32][ROWS][64 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int hd = p.hd;

    {
        constexpr int C8 = HDP / 8;      // 16-byte chunks per row
        const unsigned short* kb = p.kb + (long)b * p.s_k * p.ldk + (long)h * hd;
        const unsigned short* vb = p.vb + (long)b * p.s_k * p.ldv + (long)h * hd;
        for (int i = tid; i < ROWS * C8; i += NW * 64) {
            const int row = i / C8, c = i % C8;
            bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row < p.s_k && c * 8 < hd) {
                kv = *reinterpret_cast<const bf16x8*>(kb + (long)row * p.ldk + c * 8);
                vv = *reinterpret_cast<const bf16x8*>(vb + (long)row * p.ldv + c * 8);
            }
            *reinterpret_cast<bf16x8*>(Ks + row * KSB + c * 16) = kv;
            *reinterpret_cast<bf16x8*>(Vs + ((c >> 2) * ROWS + row) * 64 + (c & 3) * 16) = vv;
        }
    }
    __syncthreads();

    // transposing-read lane address inside a [4 keys][32 columns] block: lane 4q+p of a 16-lane group -> row q, columns 4p..4p+3
    const int tr_off = ((lane & 15) >> 2) * 64 + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2 + lh * 4 * 64;

    const int nqt = (p.s_q + 31) / 32;
    for (int qt = wave; qt < nqt; qt += NW) {
        const int qi = qt * 32 + l31;
        const int qrow = qi < p.s_q ? qi : p.s_q - 1;
        bf16x8 qf[HDP / 16];
        {
            const unsigned short* qb = p.qb + ((long)b * p.s_q + qrow) * p.ldq + (long)h * hd;
#pragma unroll
            for (int kk = 0; kk < HDP / 16; ++kk) {
                const int d = kk * 16 + 8 * lh;
                bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
                if (d < hd) t = *reinterpret_cast<const bf16x8*>(qb + d);
                qf[kk] = t;
            }
        }
        f32x16 o[HDP / 32];
#pragma unroll
        for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
        float m = -INFINITY, sum = 0.0f;
        const int nt = CAUSAL ? (qt + 1 < NT ? qt + 1 : NT) : NT;
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < HDP / 16; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 32 + l31) * KSB + (kk * 16 + 8 * lh) * 2);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st, 0, 0, 0);
            }
            float mt = -INFINITY;
            if (CAUSAL || (t + 1) * 32 > p.s_k) {            // only the last key tile (or a causal one) has keys to mask: wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool ok = key < p.s_k && (!CAUSAL || key <= qi);
                    st[r] = ok ? st[r] * p.scale : -INFINITY;
                    mt = fmaxf(mt, st[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[r] *= p.scale;
                    mt = fmaxf(mt, st[r]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m, mt);
            const float alpha = __expf(m - m_new);
            float ps = 0.0f;
            bf16x8 pf[2];
            {
                unsigned pw[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float e0 = __expf(st[r] - m_new), e1 = __expf(st[r + 1] - m_new);
                    ps += e0;                              // the normaliser sums the un-rounded weights (same order as one by one)
                    ps += e1;
                    pw[r >> 1] = f32x2_to_bf16x2_bits(e0, e1);
                }
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                pf[0] = __builtin_bit_cast(bf16x8, u32x4_t{pw[0], pw[1], pw[2], pw[3]});
                pf[1] = __builtin_bit_cast(bf16x8, u32x4_t{pw[4], pw[5], pw[6], pw[7]});
            }
            sum = sum * alpha + ps;
            m = m_new;
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* vblk = Vs + (db * ROWS + t * 32 + 16 * s) * 64 + tr_off;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + 8 * 64));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], o[db], 0, 0, 0);
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        if (p.out_q8) {
            // block-scaled fp8 output (the out-projection's MX operand): the two lanes of a query (halves 0 / 1) hold the 32 values
            // of a d block between them -- block maximum, E8M0 byte, 16 e4m3fn bytes per lane (hd % 32 == 0 on this path)
            const long orow = (long)b * p.s_q + qi;
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db) {
                float am = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(o[db][r] * inv));
                am = fmaxf(am, __shfl_xor(am, 32));
                const unsigned e8 = mx_scale_byte(am);
                const float qs = mx_inv_scale(e8);
                if (qi < p.s_q && db * 32 < hd) {
                    unsigned char* o8 = p.out_q8 + orow * p.ldo + (long)h * hd + db * 32 + 4 * lh;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<unsigned*>(o8 + 8 * g) = pack4_fp8(o[db][4 * g] * inv * qs, o[db][4 * g + 1] * inv * qs,
                                                                             o[db][4 * g + 2] * inv * qs, o[db][4 * g + 3] * inv * qs);
                    if (lh == 0) p.out_scales[mx_scale_offset(orow, (h * hd + db * 32) >> 5, p.out_srows)] = (unsigned char)e8;
                }
            }
        } else if (qi < p.s_q) {
            unsigned short* ob = p.out_b + ((long)b * p.s_q + qi) * p.ldo + (long)h * hd;
#pragma unroll
            for (int db = 0; db < HDP / 32; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * lh;
                    if (d < hd) {
                        ushort4 tt;
                        tt.x = f32_to_bf16_bits(o[db][4 * g] * inv); tt.y = f32_to_bf16_bits(o[db][4 * g + 1] * inv);
                        tt.z = f32_to_bf16_bits(o[db][4 * g + 2] * inv); tt.w = f32_to_bf16_bits(o[db][4 * g + 3] * inv);
                        *reinterpret_cast<ushort4*>(ob + d) = tt;
                    }
                }
        }
    }
}

template <int HDP, int NT, bool CAUSAL>
static hipError_t launch_inst_b(const AttnParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)NT * 32 * (HDP * 2 + 16 + HDP * 2);
    static bool attr_set = false;
    constexpr int NW = NT >= 7 ? 8 : 4;
    auto kern = attn_bf16_kernel<HDP, NT, CAUSAL, NW>;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    FERN_LAUNCH(kern, dim3(p.batch * p.heads), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

template <int HDP>
static hipError_t launch_hd_b(const AttnParams& p, hipStream_t s) {
    const int nt = (p.s_k + 31) / 32;
    if (p.causal) {
        if (nt <= 1) return launch_inst_b<HDP, 1, true>(p, s);
        if (nt <= 3) return launch_inst_b<HDP, 3, true>(p, s);
        return hipErrorInvalidValue;
    }
    if (nt <= 1) return launch_inst_b<HDP, 1, false>(p, s);
    if (nt <= 3) return launch_inst_b<HDP, 3, false>(p, s);
    if (nt <= 7) return launch_inst_b<HDP, 7, false>(p, s);
    return hipErrorInvalidValue;
}

template <int HDP, int NT, bool CAUSAL>
static hipError_t launch_inst(const AttnParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)NT * 32 * (HDP + 4 + HDP) * sizeof(float);
    static bool attr_set = false;
    // 197-token ViT heads: 7 query tiles -> 8 waves (two per SIMD) so one wave's softmax VALU work overlaps its partner's MFMAs
    constexpr int NW = NT >= 7 ? 8 : 4;
    auto kern = attn_f32_kernel<HDP, NT, CAUSAL, NW>;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    FERN_LAUNCH(kern, dim3(p.batch * p.heads), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

template <int HDP>
static hipError_t launch_hd(const AttnParams& p, hipStream_t s) {
    const int nt = (p.s_k + 31) / 32;
    if (p.causal) {
        if (nt <= 1) return launch_inst<HDP, 1, true>(p, s);
        if (nt <= 3) return launch_inst<HDP, 3, true>(p, s);
        return hipErrorInvalidValue;
    }
    if (nt <= 1) return launch_inst<HDP, 1, false>(p, s);
    if (nt <= 3) return launch_inst<HDP, 3, false>(p, s);
    if (nt <= 7) {
        static const bool chunked = [] { const char* e = getenv("FERN_ATTN_CHUNKED"); return !(e && e[0] == '0'); }();      // A/B switch
        if (chunked && nt > 4 && p.s_q <= 256 && !p.out_b) {      // the key-chunked form: every wave owns one query tile
            constexpr size_t lds = (size_t)4 * 32 * (HDP + 4 + HDP) * sizeof(float);
            static bool attr_set = false;
            auto kern = attn_f32_chunked_kernel<HDP, 7, 8, 2>;
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                attr_set = true;
            }
            FERN_LAUNCH(kern, dim3(p.batch * p.heads), dim3(8 * 64), lds, s, p);
            return hipGetLastError();
        }
        return launch_inst<HDP, 7, false>(p, s);
    }
    return hipErrorInvalidValue;
}

// ---- ONE query per (batch, head) (the class token of the last ViT block: clip_block_cls_only) ------------------------------------------
// The tiled kernels stage the head's whole K / V image in LDS and then use one query row of one MFMA tile: 47.9 us for 64 x 12 heads of
// 197 keys (77 MB of fp32 K / V at 1.6 TB/s).  Here a workgroup (4 waves) reads K and V straight from memory, lane = head dimension:
// wave w takes keys w, w + 4, ...; a key's score is a 64-lane product + xor-tree sum, its weight exp(score - max) over all keys, the output
// sum_j weight_j v_j[d] per lane, the four waves' partial outputs added in wave order.  UNR row loads stay in flight per wave.
constexpr int SQ1_MAX_KEYS = 1024;
template <int UNR>
__global__ __launch_bounds__(256) void attn_f32_single_query_kernel(AttnParams p) {
    __shared__ float sc[SQ1_MAX_KEYS];
    __shared__ float red[4][64];
    __shared__ float wred[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads, hd = p.hd;
    const int lc = lane < hd ? lane : 0;                                     // clamped: every load is unconditional
    const float* kb = p.k + (long)b * p.s_k * p.ldk + (long)h * hd + lc;
    const float* vb = p.v + (long)b * p.s_k * p.ldv + (long)h * hd + lc;
    const float qd = lane < hd ? p.q[(long)b * p.ldq + (long)h * hd + lane] * p.scale : 0.0f;      // s_q == 1
    for (int j0 = wave; j0 < p.s_k; j0 += 4 * UNR) {
        float kv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = j0 + 4 * u;
            kv[u] = kb[(long)(j < p.s_k ? j : p.s_k - 1) * p.ldk];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float sdot = qd * kv[u];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) sdot += __shfl_xor(sdot, m);
            const int j = j0 + 4 * u;
            if (lane == 0 && j < p.s_k) sc[j] = sdot;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < p.s_k; j += 256) mx = fmaxf(mx, sc[j]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    if (lane == 0) wred[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    float sum = 0.0f;
    for (int j = tid; j < p.s_k; j += 256) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    if (lane == 0) wred[4 + wave] = sum;
    __syncthreads();
    sum = (wred[4] + wred[5]) + (wred[6] + wred[7]);
    float acc = 0.0f;
    for (int j0 = wave; j0 < p.s_k; j0 += 4 * UNR) {
        float vv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = j0 + 4 * u;
            vv[u] = vb[(long)(j < p.s_k ? j : p.s_k - 1) * p.ldv];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = j0 + 4 * u;
            acc = __builtin_fmaf(j < p.s_k ? sc[j] : 0.0f, vv[u], acc);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < hd)
        p.out[(long)b * p.ldo + (long)h * hd + lane] = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) * (1.0f / sum);
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.batch <= 0 || p.heads <= 0 || p.s_q <= 0 || p.s_k <= 0) return hipErrorInvalidValue;
    if (p.qb || p.kb || p.vb) {
        if (!p.qb || !p.kb || !p.vb || (!p.out_b && !p.out_q8) || (p.out_q8 && (!p.out_scales || (p.hd & 31))) || (p.hd & 7) || (p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3)) return hipErrorInvalidValue;
        if (p.causal && p.s_q != p.s_k) return hipErrorInvalidValue;
        if (p.hd <= 32) return launch_hd_b<32>(p, s);
        if (p.hd <= 64) return launch_hd_b<64>(p, s);
        if (p.hd <= 96) return launch_hd_b<96>(p, s);
        return hipErrorInvalidValue;
    }
    if ((p.hd & 3) || (p.ldq & 3) || (p.ldk & 3) || (p.ldv & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
    if (p.causal && p.s_q != p.s_k) return hipErrorInvalidValue;
    if (p.s_q == 1 && !p.causal && p.hd <= 64 && p.s_k <= SQ1_MAX_KEYS && !p.out_b) {      // one query per head: no K / V image, no MFMA tile
        FERN_LAUNCH(attn_f32_single_query_kernel<25>, dim3(p.batch * p.heads), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (p.hd <= 32) return launch_hd<32>(p, s);
    if (p.hd <= 64) return launch_hd<64>(p, s);
    if (p.hd <= 96) return launch_hd<96>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace fern
