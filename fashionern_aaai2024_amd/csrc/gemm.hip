// fp32 MFMA GEMM for gfx950 (CDNA4): C[M,N] = A[M,K] * W[N,K]^T with fused epilogues.
//
// This is the workhorse of the encode -> fuse -> rank path: it replaces every nn.Linear /
// Conv2d(patch-embed) / cosine-similarity matmul the reference issues through cuBLAS/cuDNN
// (SURVEY.md 2.2 rows K1-K6, K8).  The reference evaluates in fp32, so the matrix cores are used
// through v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 FLOP/clk/SIMD = 157 TFLOP/s peak).
//
// Tiling (wave64, 4 waves per workgroup):
//   * block tile BM x BN x 32, each wave owns a (BM/WAVES_M) x (BN/WAVES_N) sub-tile made of 32x32 MFMA tiles;
//   * A and W tiles are staged global -> registers -> LDS (16-byte loads, issue-early / write-late so the
//     next tile's HBM/L2 latency hides under the current tile's ~4k MFMA cycles);
//   * LDS rows are padded to 36 floats: every ds_read_b128 of an operand fragment is conflict-free
//     (16-lane groups hit 16 distinct 4-bank slots) and stays 16-byte aligned;
//   * the k index inside a 32-wide tile is permuted (lane half h reads k = 8*kk + 4*h + j) so that one
//     ds_read_b128 feeds four MFMAs; both operands use the same permutation, so the sum is unchanged;
//   * workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of tiles.
#include "kernels.h"

namespace fern {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_S = BK + 4;   // padded row stride (floats)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AJ = BM / 32, WJ = BN / 32;   // float4 loads per thread per tile

    __shared__ __attribute__((aligned(16))) float As[BM * LDS_S];
    __shared__ __attribute__((aligned(16))) float Ws[BN * LDS_S];
    __shared__ float red[WAVES_N][BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- XCD-aware, bijective workgroup -> tile map (n fastest inside an XCD's contiguous run) ----
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int bm = swz / nbn, bn = swz % nbn;

    // ---- per-thread staging coordinates ----
    const int c4 = tid & 7;      // float4 column inside the 32-wide k tile
    const int r0 = tid >> 3;     // 0..31
    const float* a_base[AJ];
    const float* w_base[WJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int row = bm * BM + r0 + 32 * j;
        row = row < p.M ? row : p.M - 1;
        if (p.aload == ALOAD_IM2COL) {
            const int g2 = p.grid * p.grid;
            const int b = row / g2, pr = row % g2;
            const int py = pr / p.grid, px = pr % p.grid;
            a_base[j] = p.A + ((long)b * 3 * p.img + (long)py * p.patch) * p.img + (long)px * p.patch;
        } else {
            a_base[j] = p.A + (long)row * p.lda;
        }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        int row = bn * BN + r0 + 32 * j;
        row = row < p.N ? row : p.N - 1;
        w_base[j] = p.W + (long)row * p.ldw;
    }

    f32x4 a_stage[AJ], w_stage[WJ];
    auto stage_load = [&](int k0) {
        const int k = k0 + c4 * 4;
        long a_off = k;
        if (p.aload == ALOAD_IM2COL) {
            const int pp = p.patch * p.patch;
            const int c = k / pp, rem = k % pp;
            const int ky = rem / p.patch, kx = rem % p.patch;
            a_off = ((long)c * p.img + ky) * p.img + kx;
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) a_stage[j] = *reinterpret_cast<const f32x4*>(a_base[j] + a_off);
#pragma unroll
        for (int j = 0; j < WJ; ++j) w_stage[j] = *reinterpret_cast<const f32x4*>(w_base[j] + k);
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * j) * LDS_S + c4 * 4]) = a_stage[j];
#pragma unroll
        for (int j = 0; j < WJ; ++j) *reinterpret_cast<f32x4*>(&Ws[(r0 + 32 * j) * LDS_S + c4 * 4]) = w_stage[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.K / BK;
    stage_load(0);
    stage_write();
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage_load((kt + 1) * BK);   // in flight during the MFMA phase below
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[(wm * WM + i * 32 + l31) * LDS_S + kk * 8 + 4 * lh]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Ws[(wn * WN + j * 32 + l31) * LDS_S + kk * 8 + 4 * lh]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            stage_write();
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const int row_w = bm * BM + wm * WM;
    const int col_w = bn * BN + wn * WN;
    if (p.epi < EPI_RELU_DOT) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col_w + j * 32 + l31;
                if (col >= p.N) continue;
                const float bia = p.bias ? p.bias[col] : 0.0f;
                float sc = 1.0f, sh = 0.0f;
                if (p.epi == EPI_COLAFFINE_TANH) { sc = p.aux0[col]; sh = p.aux1[col]; }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row >= p.M) continue;
                    float v = acc[i][j][r] + bia;
                    long orow = row;
                    switch (p.epi) {
                        case EPI_BIAS_GELU: v = gelu_erf(v); break;
                        case EPI_BIAS_RELU: v = fmaxf(v, 0.0f); break;
                        case EPI_BIAS_RESIDUAL: v += p.R[(long)row * p.ldc + col]; break;
                        case EPI_COLAFFINE_TANH: v = tanhf(v * sc + sh); break;
                        case EPI_PATCH_EMBED: {
                            const int g2 = p.grid * p.grid;
                            orow = row + row / g2 + 1;
                            v += p.aux0[(long)((row % g2) + 1) * p.N + col];
                        } break;
                        default: break;
                    }
                    p.C[orow * p.ldc + col] = v;
                }
            }
    } else {
        // reduce epilogues: one partial sum per (row, column block); fixed summation order => deterministic
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // row inside the block tile
                const int row = bm * BM + rl;
                const int rowc = row < p.M ? row : p.M - 1;
                float s = 0.0f;
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
                    if (col < p.N) {
                        float v = acc[i][j][r] + p.bias[col];
                        if (p.epi == EPI_RELU_DOT) v = fmaxf(v, 0.0f) * p.aux0[col];
                        else v = tanhf((v - mu) * inv + beta) * grow[col] * p.aux0[col];
                        s += v;
                    }
                }
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 8);
                s += __shfl_xor(s, 4);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 1);
                if (l31 == 0) red[wn][rl] = s;
            }
        }
        __syncthreads();
        if (tid < BM) {
            const int row = bm * BM + tid;
            if (row < p.M) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < WAVES_N; ++w) s += red[w][tid];
                p.partial[(long)row * nbn + bn] = s;
            }
        }
    }
}

struct TileCfg { int bm, bn; float eff; };
static const TileCfg kCfgs[4] = {{128, 128, 1.00f}, {64, 128, 0.93f}, {128, 64, 0.93f}, {64, 64, 0.86f}};

static int choose_cfg(int M, int N) {
    int best = 0;
    double best_cost = 1e300;
    for (int c = 0; c < 4; ++c) {
        const long nb = (long)((M + kCfgs[c].bm - 1) / kCfgs[c].bm) * ((N + kCfgs[c].bn - 1) / kCfgs[c].bn);
        const long rounds = (nb + 255) / 256;
        const double cost = (double)rounds * kCfgs[c].bm * kCfgs[c].bn / kCfgs[c].eff;
        if (cost < best_cost * 0.999) { best_cost = cost; best = c; }
    }
    return best;
}

int gemm_num_col_blocks(int M, int N) {
    const int c = choose_cfg(M, N);
    return (N + kCfgs[c].bn - 1) / kCfgs[c].bn;
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K <= 0 || (p.K % BK) != 0 || (p.ldw & 3) || (p.aload == ALOAD_PLAIN && (p.lda & 3))) return hipErrorInvalidValue;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return hipErrorInvalidValue;
    if (p.aload == ALOAD_IM2COL && ((p.patch & 3) || (p.img & 3))) return hipErrorInvalidValue;
    const int c = choose_cfg(p.M, p.N);
    const int nb = ((p.M + kCfgs[c].bm - 1) / kCfgs[c].bm) * ((p.N + kCfgs[c].bn - 1) / kCfgs[c].bn);
    switch (c) {
        case 0: hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 64, 64>), dim3(nb), dim3(256), 0, s, p); break;
        case 1: hipLaunchKernelGGL((gemm_f32_kernel<64, 128, 32, 64>), dim3(nb), dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((gemm_f32_kernel<128, 64, 64, 32>), dim3(nb), dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 32, 32>), dim3(nb), dim3(256), 0, s, p); break;
    }
    return hipGetLastError();
}

}  // namespace fern
