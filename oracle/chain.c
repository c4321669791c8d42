/*
 * oracle/chain.c -- TEST INFRASTRUCTURE ONLY (never linked into libfern.so, never called by the product path).
 *
 * Bit-exact CPU restatement of the fp32 cosine sweep of the HIP path: scores[b][n] = q[b] . g[n] accumulated as ONE
 * sequential fp32 fused-multiply-add chain in the k order the MFMA kernels use.
 *
 * What it follows.  The reference computes `1 - predicted_features @ index_features.T` in fp32
 * (/root/reference/run/test/test_fiq.py:49) and leaves the summation order to the BLAS; the HIP path fixes one:
 * v_mfma_f32_32x32x2_f32 is, per output element, fma(a[k1], b[k1], fma(a[k0], b[k0], c)) (one rounding per product,
 * verified bitwise by tools/probe/mfma16_probe.hip), and every tile configuration of fashionern_aaai2024_amd/csrc/gemm.hip
 * feeds, inside each group of 8 consecutive k, MFMA e = 0..3 with k0 = 8g + e (lane half 0) and k1 = 8g + 4 + e (lane
 * half 1).  The chain order is therefore 8g, 8g+4, 8g+1, 8g+5, 8g+2, 8g+6, 8g+3, 8g+7 for g = 0, 1, ...
 *
 * With this order restated, "identical top-K ordering" can be asserted bit for bit on random unit rows, not only on
 * operands whose dot products are exact in any order.  Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/chain.py).
 */
#include <math.h>
#include <stdint.h>

/* scores [B][N] (row-major) from q [B][D], g [N][D]; D % 8 == 0.  fmaf() is the correctly rounded C99 fused multiply-add. */
void fern_oracle_chain_scores(const float* q, const float* g, float* scores, int64_t B, int64_t N, int64_t D) {
    for (int64_t b = 0; b < B; ++b) {
        const float* qr = q + b * D;
        for (int64_t n = 0; n < N; ++n) {
            const float* gr = g + n * D;
            float acc = 0.0f;
            for (int64_t k8 = 0; k8 < D; k8 += 8) {
                for (int e = 0; e < 4; ++e) {
                    acc = fmaf(qr[k8 + e], gr[k8 + e], acc);
                    acc = fmaf(qr[k8 + 4 + e], gr[k8 + 4 + e], acc);
                }
            }
            scores[b * N + n] = acc;
        }
    }
}
