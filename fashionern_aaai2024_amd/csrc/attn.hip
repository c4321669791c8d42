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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bool ok = key < p.s_k && (!CAUSAL || key <= qi);
                st[r] = ok ? st[r] : -INFINITY;
                mt = fmaxf(mt, st[r]);
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
    hipLaunchKernelGGL(kern, dim3(p.batch * p.heads), dim3(NW * 64), lds, s, p);
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
    if (nt <= 7) return launch_inst<HDP, 7, false>(p, s);
    return hipErrorInvalidValue;
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.batch <= 0 || p.heads <= 0 || p.s_q <= 0 || p.s_k <= 0) return hipErrorInvalidValue;
    if ((p.hd & 3) || (p.ldq & 3) || (p.ldk & 3) || (p.ldv & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
    if (p.causal && p.s_q != p.s_k) return hipErrorInvalidValue;
    if (p.hd <= 32) return launch_hd<32>(p, s);
    if (p.hd <= 64) return launch_hd<64>(p, s);
    if (p.hd <= 96) return launch_hd<96>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace fern
