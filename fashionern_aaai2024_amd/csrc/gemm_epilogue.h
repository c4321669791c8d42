// Shared GEMM epilogue (bias / activation / residual / reduce forms) for the fp32 and bf16 MFMA GEMM kernels: both
// v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 leave the 32x32 fp32 accumulator in the same register layout.
#pragma once
#include "kernels.h"

namespace fern {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)), with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. at
// fp32 rounding level) evaluated branch-free: 1 + erf(z) = 2 - P(t) e^{-z^2} for z >= 0 and P(t) e^{-z^2} for z < 0
// (t = 1 / (1 + p |z|)), which also avoids the cancellation of 1 + erf(z) in the negative tail.  ~15 VALU ops per
// element instead of the ~45 of libm's erff: the GELU epilogue of the 3072-wide MLP GEMMs was VALU-bound.
// Evaluated on two elements at a time: the polynomial and the products compile to packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32), the
// reciprocal is the hardware's 1-ulp v_rcp_f32 rather than a correctly rounded division (a 6e-8 relative change of t, below the
// formula's own 1.5e-7) and the exponential is v_exp_f32 on a pre-scaled argument: ~17 issue slots per element (the scalar form with
// a correctly rounded division took ~30, libm's erff ~45).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 az = __builtin_elementwise_abs(z);
    const f32x2 d = az * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    const f32x2 m = az * az * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
    const f32x2 pe = poly * t * e;
    const f32x2 cdf2 = {z[0] >= 0.0f ? 2.0f - pe[0] : pe[0], z[1] >= 0.0f ? 2.0f - pe[1] : pe[1]};
    return x * 0.5f * cdf2;
}

// GELU of the reduced-precision GEMM family (bf16 / fp8 / block-scaled fp8 operands: outputs are rounded to bf16 or fp8, or feed a
// residual stream that already carries operand rounding): erf from Abramowitz & Stegun 7.1.28,
//   1 - erf(z) = (1 + a1 z + ... + a6 z^6)^-16,  |error| <= 3e-7,
// i.e. ONE reciprocal and no exponential; 1 + erf(z) for z < 0 is the same power directly (no cancellation in the tail).  Written on
// float2 so that the polynomial and the four squarings compile to packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32): ~11 issue slots
// per element against ~30 for gelu_erf (two quarter-rate transcendentals and a correctly rounded division) -- the GELU epilogue,
// not the MFMA loop, was the longest phase of the block-scaled c_fc GEMM.  |gelu_fast - exact| <= 8.2e-7 over [-12, 12] (fp32).
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 az = __builtin_elementwise_abs(z);
    f32x2 p = az * 0.0000430638f + 0.0002765672f;
    p = p * az + 0.0001520143f;
    p = p * az + 0.0092705272f;
    p = p * az + 0.0422820123f;
    p = p * az + 0.0705230784f;
    p = p * az + 1.0f;
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;                                              // may overflow to +inf for |x| > ~19: 1 / inf = 0 is the right limit
    const f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    const f32x2 cdf2 = {z[0] >= 0.0f ? 2.0f - r[0] : r[0], z[1] >= 0.0f ? 2.0f - r[1] : r[1]};
    return x * 0.5f * cdf2;
}

// GELU of the BLOCK-SCALED fp8 GEMM family (outputs are e4m3 bytes, or feed operands that will be): the tanh form written as
// x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3) -- one exp2 and one reciprocal per element, ~6.5 issue slots against ~12.5 for
// gelu_fast2.  |gelu_tanh - exact| <= 4.8e-4 (at |x| ~ 2.2): far below the 2^-4 relative step of an e4m3 value, NOT below fp32 or bf16
// rounding, so no other family uses it.  The c_fc launch of that mode is bound by its epilogue's vector instructions (tools/probe/
// mx_lab.hip: 9.0 -> 6.8 us per 256x256 tile, the launch -10 %).  exp2 overflow (x << 0) gives 1/inf = 0 and underflow x * 1: both limits.
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    const f32x2 x2 = x * x;
    const f32x2 t = x2 * (-2.0f * 0.7978845608f * 0.044715f * 1.4426950409f) + (-2.0f * 0.7978845608f * 1.4426950409f);
    const f32x2 w = t * x;                                   // -2u log2(e)
    const f32x2 e = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    const f32x2 d = e + 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
}
// FAST (template parameter of the epilogues below): 0 = gelu_erf2 (fp32 parity mode), 1 = gelu_fast2 (bf16 / per-row fp8 families),
// 2 = gelu_tanh2 (block-scaled fp8 family)
template <int FAST>
__device__ __forceinline__ f32x2 gelu_of(f32x2 x) {
    if (FAST == 2) return gelu_tanh2(x);
    if (FAST == 1) return gelu_fast2(x);
    return gelu_erf2(x);
}

// One 32x32 accumulator tile.  GUARD = false: the tile lies fully inside the matrix (wave-uniform test by the caller), so
// there is no per-element predicate at all -- the 16 residual / scale loads issue back to back behind ONE wait, and so do
// the 16 stores.  (With a predicate per element every load sits in its own exec-masked block and the compiler waits for
// it -- and for every store before it -- on the spot: 16 serial memory round trips per tile.)
template <int EPI, bool OUT_BF16, bool SCALED, bool GUARD, int FAST = 0>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, const f32x16& acc, int row0, int col0, int l31, int lrow, int loff) {
    constexpr bool resid = EPI == EPI_BIAS_RESIDUAL || EPI == EPI_BIAS_RESIDUAL_RELU;
    const int col = col0 + l31;
    if (GUARD && col >= p.N) return;
    const float bia = p.bias ? p.bias[col] : 0.0f;
    const float qw = SCALED ? p.scale_w[col] : 1.0f;          // fp8 operands: per-output-channel weight scale
    const float* qa = SCALED ? p.scale_a + row0 : nullptr;    // ... and per-row activation scales
    float sc = 1.0f, sh = 0.0f;
    if (EPI == EPI_COLAFFINE_TANH) { sc = p.aux0[col]; sh = p.aux1[col]; }
    if (EPI == EPI_PATCH_EMBED) {
        const int g2 = p.grid * p.grid;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + lrow;
            if (!GUARD || row < p.M) {
                const long orow = row + row / g2 + 1;
                p.C[orow * p.ldc + col] = acc[r] + bia + p.aux0[(long)((row % g2) + 1) * p.N + col];
            }
        }
        return;
    }
    if constexpr (!GUARD) {
        // Full tile (round 4): BUFFER addressing.  A tile's 16 accesses differ by a wave-uniform row offset only, so the address is
        // (descriptor of the tile's first element, SGPRs) + (one lane offset, ONE VGPR for the whole kernel) + (row offset, an SGPR):
        // no vector address arithmetic at all.  With flat global addressing hipcc kept a 64-bit address per element and advanced it
        // with a v_lshl_add_u64 per access -- once for the residual loads and once more for the stores (R may alias C) -- a third of
        // the epilogue's vector instructions and 32 live registers.  Same loads, same stores, same order, same values.
        constexpr unsigned kRsrcFlags = 0x00020000u;      // gfx9 buffer descriptor DWORD3: 32-bit raw, no swizzle
        constexpr int ES = OUT_BF16 ? 2 : 4;
        const int voff = loff * ES;                                       // lane part: (4 lh) rows + column l31
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(p.C) + ((long)row0 * p.ldc + col0) * ES, 0, 0x7FFFFFFF, kRsrcFlags);
        float add[16], qs[16];
        if (resid) {
            const char* rb = OUT_BF16 ? reinterpret_cast<const char*>(p.Rb) : reinterpret_cast<const char*>(p.R);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rb) + ((long)row0 * p.ldc + col0) * ES, 0, 0x7FFFFFFF, kRsrcFlags);
#pragma unroll
            for (int r = 0; r < 16; ++r) {      // all sixteen addends before the first store (in place: R == C)
                const int soff = ((r & 3) + 8 * (r >> 2)) * (int)p.ldc * ES;
                if (OUT_BF16) add[r] = bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(rr, voff, soff, 0));
                else add[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, voff, soff, 0));
            }
        }
        if (SCALED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) qs[r] = qa[(r & 3) + 8 * (r >> 2) + lrow] * qw;
        }
        auto put = [&](int r, float v) {
            if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.0f);
            else if (EPI == EPI_BIAS_RESIDUAL) v += add[r];
            else if (EPI == EPI_BIAS_RESIDUAL_RELU) v = fmaxf(v + add[r], 0.0f);
            else if (EPI == EPI_COLAFFINE_TANH) v = tanhf(v * sc + sh);
            const int soff = ((r & 3) + 8 * (r >> 2)) * (int)p.ldc * ES;
            if (OUT_BF16) __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16_bits(v), rc, voff, soff, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff, soff, 0);
        };
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v2 = {SCALED ? acc[r] * qs[r] + bia : acc[r] + bia, SCALED ? acc[r + 1] * qs[r + 1] + bia : acc[r + 1] + bia};
                const f32x2 g2 = gelu_of<FAST>(v2);
                put(r, g2[0]);
                put(r + 1, g2[1]);
                if (FAST == 2) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) put(r, SCALED ? acc[r] * qs[r] + bia : acc[r] + bia);
        }
        return;
    }
    // The residual stream is updated in place (R == C): the compiler keeps every load behind the preceding store, so the
    // memory addends of a tile are fetched together BEFORE its first store.
    const float* Rt = resid && !OUT_BF16 ? p.R + (long)row0 * p.ldc + col0 : nullptr;
    const unsigned short* Rbt = resid && OUT_BF16 ? p.Rb + (long)row0 * p.ldc + col0 : nullptr;      // bf16 residual stream (block-scaled family)
    float add[16], qs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ru = (r & 3) + 8 * (r >> 2);
        const bool ok = !GUARD || row0 + ru + lrow < p.M;
        if (resid && OUT_BF16) add[r] = ok ? bf16_bits_to_f32(Rbt[(long)ru * p.ldc + loff]) : 0.0f;
        else if (resid) add[r] = ok ? Rt[(long)ru * p.ldc + loff] : 0.0f;
        if (SCALED) qs[r] = ok ? qa[ru + lrow] * qw : 0.0f;
    }
    float* Ct = OUT_BF16 ? nullptr : p.C + (long)row0 * p.ldc + col0;
    unsigned short* Cb = OUT_BF16 ? reinterpret_cast<unsigned short*>(p.C) + (long)row0 * p.ldc + col0 : nullptr;
    auto emit = [&](int r, float v) {                       // activation applied; residual / store of register r
        const int ru = (r & 3) + 8 * (r >> 2);
        if (!GUARD || row0 + ru + lrow < p.M) {
            if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.0f);
            else if (EPI == EPI_BIAS_RESIDUAL) v += add[r];
            else if (EPI == EPI_BIAS_RESIDUAL_RELU) v = fmaxf(v + add[r], 0.0f);
            else if (EPI == EPI_COLAFFINE_TANH) v = tanhf(v * sc + sh);
            if (OUT_BF16) Cb[(long)ru * p.ldc + loff] = f32_to_bf16_bits(v);
            else Ct[(long)ru * p.ldc + loff] = v;
        }
    };
    if (EPI == EPI_BIAS_GELU) {                              // two elements per packed instruction, stored as they are produced
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 v2 = {SCALED ? acc[r] * qs[r] + bia : acc[r] + bia, SCALED ? acc[r + 1] * qs[r + 1] + bia : acc[r + 1] + bia};
            const f32x2 g2 = gelu_of<FAST>(v2);
            emit(r, g2[0]);
            emit(r + 1, g2[1]);
            if (FAST == 2) __builtin_amdgcn_sched_barrier(0);      // one pair at a time: interleaved pairs of the two-transcendental form spill at 128 VGPRs
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) emit(r, SCALED ? acc[r] * qs[r] + bia : acc[r] + bia);
    }
}

// Plain (element-wise) epilogue of one epilogue kind: the kind is a template parameter so that the switch is taken once per
// kernel, not once per element, and the tile loops stay small enough to unroll fully (register-indexed accumulators).
// C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// An element's address = (wave-uniform tile / register-row part, kept in SGPRs) + (lane part: 4 * half rows + column):
// one 32-bit VGPR offset serves all 16 loads and stores of a tile (ldc < 2^24 on this path, so the lane part fits an int).
template <int EPI, bool OUT_BF16, bool SCALED, int FAST = 0, int TM, int TN>
__device__ __forceinline__ void plain_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], int row_w, int col_w, int l31, int lh) {
    const int lrow = 4 * lh;
    const int loff = lrow * (int)p.ldc + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row0 = row_w + i * 32, col0 = col_w + j * 32;       // wave-uniform
            if (row0 + 32 <= p.M && col0 + 32 <= p.N) epilogue_tile<EPI, OUT_BF16, SCALED, false, FAST>(p, acc[i][j], row0, col0, l31, lrow, loff);
            else if (row0 < p.M && col0 < p.N) epilogue_tile<EPI, OUT_BF16, SCALED, true, FAST>(p, acc[i][j], row0, col0, l31, lrow, loff);
            // one 32x32 tile at a time: without this fence the scheduler hoists the address arithmetic and the loads of all
            // TM x TN tiles to the top and the kernel's register budget (= its occupancy) is set by the epilogue
            __builtin_amdgcn_sched_barrier(0);
        }
}

// The cosine sweep of the fused top-K (rows = queries, columns = gallery rows): no score is stored.  Each 32x32 tile is
// compared with its 16 queries' bounds (kernels.h: topk_filter_tile); thr_key == null rejects everything (tuner trials).
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void filter_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], int bm, int bn, int wm, int wn, int l31, int lh) {
    if (!p.filt.thr_key) return;
    const int row_w = bm * BM + wm * WM;
    const int col_w = bn * BN + wn * WN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q0 = row_w + i * 32;
        if (q0 >= p.M) continue;
        float bound[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            bound[r] = qi < p.M ? filter_bound(p.filt.thr_key[qi]) : __builtin_inff();
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long n = col_w + j * 32 + l31;
            topk_filter_tile(acc[i][j], bound, q0, lh, n, n < p.N, p.M, p.filt);
            __builtin_amdgcn_sched_barrier(0);      // one tile at a time (see plain_epilogue)
        }
    }
}

// Shared epilogue of the fp32 and bf16 GEMM kernels (ALLOW_BF16_OUT: only the bf16 kernel stores bf16 outputs).
template <int BM, int BN, int WM, int WN, int TM, int TN, int WAVES_N, bool ALLOW_BF16_OUT = false, int RGELU = 1 /* GELU of the reduced-precision families: 1 fast, 2 tanh */,
          bool ALLOW_SCALED = true /* per-row / per-channel fp8 scales folded back here (p.scale_a); the block-scaled family has none and leaves the code out */>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], int bm, int bn, int nbn,
                                              int wm, int wn, int l31, int lh, int tid) {
    const int row_w = bm * BM + wm * WM;
    const int col_w = bn * BN + wn * WN;
    if (!epi_is_reduce(p.epi)) {
        if (ALLOW_BF16_OUT && ALLOW_SCALED && p.scale_a) {      // fp8 operands: scales folded back here; bias / GELU / residual forms
            if (p.out_bf16) {
                if (p.epi == EPI_BIAS_GELU) plain_epilogue<EPI_BIAS_GELU, true, true, RGELU>(p, acc, row_w, col_w, l31, lh);
                else plain_epilogue<EPI_BIAS, true, true>(p, acc, row_w, col_w, l31, lh);
            } else {
                if (p.epi == EPI_BIAS_RESIDUAL) plain_epilogue<EPI_BIAS_RESIDUAL, false, true>(p, acc, row_w, col_w, l31, lh);
                else if (p.epi == EPI_BIAS_GELU) plain_epilogue<EPI_BIAS_GELU, false, true, RGELU>(p, acc, row_w, col_w, l31, lh);
                else plain_epilogue<EPI_BIAS, false, true>(p, acc, row_w, col_w, l31, lh);
            }
            return;
        }
        if (ALLOW_BF16_OUT && p.out_bf16) {
            if (!ALLOW_SCALED && p.epi == EPI_BIAS_RESIDUAL) {      // block-scaled family: bf16 residual stream, updated in place
                plain_epilogue<EPI_BIAS_RESIDUAL, true, false>(p, acc, row_w, col_w, l31, lh);
                return;
            }
            switch (p.epi) {      // bf16 outputs feed the next bf16 GEMM: bias (+ GELU / ReLU) only
                case EPI_BIAS_GELU: plain_epilogue<EPI_BIAS_GELU, true, false, RGELU>(p, acc, row_w, col_w, l31, lh); break;
                case EPI_BIAS_RELU: plain_epilogue<EPI_BIAS_RELU, true, false>(p, acc, row_w, col_w, l31, lh); break;
                default: plain_epilogue<EPI_BIAS, true, false>(p, acc, row_w, col_w, l31, lh); break;
            }
            return;
        }
        switch (p.epi) {
            case EPI_BIAS_GELU: plain_epilogue<EPI_BIAS_GELU, false, false, ALLOW_BF16_OUT ? RGELU : 0>(p, acc, row_w, col_w, l31, lh); break;
            case EPI_BIAS_RELU: plain_epilogue<EPI_BIAS_RELU, false, false>(p, acc, row_w, col_w, l31, lh); break;
            case EPI_BIAS_RESIDUAL: plain_epilogue<EPI_BIAS_RESIDUAL, false, false>(p, acc, row_w, col_w, l31, lh); break;
            case EPI_BIAS_RESIDUAL_RELU: plain_epilogue<EPI_BIAS_RESIDUAL_RELU, false, false>(p, acc, row_w, col_w, l31, lh); break;
            case EPI_COLAFFINE_TANH: plain_epilogue<EPI_COLAFFINE_TANH, false, false>(p, acc, row_w, col_w, l31, lh); break;
            case EPI_PATCH_EMBED: plain_epilogue<EPI_PATCH_EMBED, false, false>(p, acc, row_w, col_w, l31, lh); break;
            default: plain_epilogue<EPI_BIAS, false, false>(p, acc, row_w, col_w, l31, lh); break;
        }
    } else {
        // Reduce epilogues: one partial sum per (row, 32-column group) = one MFMA tile row, reduced over the 32 lanes that
        // hold its columns in a fixed xor tree and written straight from the wave (no LDS staging, no workgroup barrier).
        // The layout partial[row][N/32] does not depend on the tile configuration, so every configuration produces the same
        // partials and the finalize kernels add them in the same order: the result is independent of the tile choice and
        // of M (batch-invariant), and these GEMMs can be tuned per shape like the plain ones.
        const int ng = (p.N + 31) / 32;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int rowc = row < p.M ? row : p.M - 1;
                float mu = 0.0f, inv = 1.0f, beta = 0.0f;
                const float* grow = nullptr;
                if (p.epi == EPI_SR_LOCAL) {
                    const int pidx = rowc % 13;
                    mu = p.aux1[pidx]; inv = p.aux2[pidx]; beta = p.aux3[pidx];
                    grow = p.G + (long)(rowc / 13) * p.ldg;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col_w + j * 32 + l31;
                    float v = 0.0f;
                    if (col < p.N) {
                        v = acc[i][j][r] + p.bias[col];
                        if (p.epi == EPI_RELU_DOT) v = fmaxf(v, 0.0f) * p.aux0[col];
                        else v = tanhf((v - mu) * inv + beta) * grow[col] * p.aux0[col];
                    }
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 8);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 1);
                    const int grp = (col_w >> 5) + j;
                    if (l31 == 0 && row < p.M && grp < ng) p.partial[(long)row * ng + grp] = v;
                }
            }
        }
    }
}

}  // namespace fern
