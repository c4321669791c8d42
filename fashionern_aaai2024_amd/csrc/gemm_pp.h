// "Ping-pong" 256 x 256 GEMM tile for the reduced-precision families (bf16 / block-scaled fp8 operands), round 6.
//
// C[M,N] = A[M,K] * W[N,K]^T on v_mfma_f32_32x32x16_bf16 / v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulation, the family's fused
// epilogues (gemm_epilogue.h).  What the LDS-DMA kernels of gemm_bf16.hip could not do (flat at 0.19 / 0.23 of peak for
// four rounds) is keep the operand feed AND the MFMA pipe busy at the same time: all of a workgroup's waves walk the same
// wait -> barrier -> stage -> read -> MFMA sequence, one stage (at most two) is in flight per CU while a tile is consumed, and a
// staged tile has one compute period to arrive.  This kernel is built the other way round:
//
//   * 8 waves = 2 (M) x 4 (N), 128 x 64 outputs each (4 x 2 accumulator tiles, 128 registers), two waves per SIMD.  The two waves
//     of a SIMD belong to different M halves ("groups"), and the groups run HALF A PHASE APART: one s_barrier more at the start of
//     group 1.  Every phase is  [LDS reads]  s_barrier  [8 MFMAs under s_setprio 1]  s_barrier , so while one group's waves issue
//     their 256 cycles of MFMAs the other group's waves -- on the same SIMDs -- issue the reads of their next phase.  The phase's two
//     LDS-DMA instructions go out BETWEEN the MFMAs (VAR 1, the default: in the matrix pipe's shadow; VAR 0 issued them in the load
//     section, where they were what the OTHER group's MFMAs ended up waiting for: 965 -> 1 083 TFLOP/s at 4096^3 bf16).
//   * k tiles of 64 BYTES per row (32 bf16 / 64 fp8 k) in a ring of NB = 4 slots of 32 KiB (A rows 16 KiB + W rows 16 KiB): two
//     phases per k tile (the wave's upper / lower 64 x 64 half of C), ONE 16 KiB unit staged per phase, 48-64 KiB in flight at the
//     point where a k tile's arrival is awaited -- a unit has three to four phases (~1 us) to land instead of one.  The staging of a unit
//     follows the last read of the unit it replaces by exactly the barriers that order it (derivation below).
//   * every accumulator adds its 16-wide (bf16) / 64-wide (fp8) k groups in ascending order: bit-identical to every other tile
//     configuration of the family.
//
// Synchronisation (p = phase, k tile T = p >> 1, j = p & 1; G0 / G1 = the groups; barrier b = the b-th s_barrier after the
// prologue's; G1 executes one extra barrier first, so its phase p sits between barriers 2p .. 2p+2 and G0's between 2p-1 .. 2p+1):
//     reads:  j = 0 reads the W unit (4 x ds_read_b128) and the wave's upper A rows (4); j = 1 the lower A rows (4).
//     WAR:    W of k tile t is last read in phase 2t: G0's reads retire (lgkmcnt(0)) after barrier 4t, G1's after barrier 4t+1, and
//             every wave that has passed barrier 4t+2 knows it.  A of k tile t is last read in phase 2t+1: known after barrier 4t+4.
//             The slot of k tile t is refilled with k tile t+NB: its W unit in phase 2t+2 (issued by G0 after barrier 4t+3, by G1
//             after 4t+4), its A unit in phase 2t+3 (after barriers 4t+5 / 4t+6) -- VAR 1 issues them one barrier later still.
//     RAW:    an LDS-DMA write is ordered for a ds_read only by the ISSUING wave's counted vmcnt followed by a barrier the reader
//             has passed.  Every wave waits for k tile T+1 at the end of its load section of phase 2T+1 (all but the operations it
//             has issued after that k tile's last one: 2 x min(NB-2, nk-2-T) units, one fewer under VAR 1, whose A unit of the
//             newest k tile leaves after this wait), i.e. before barrier 4T+2 (G0) / 4T+3 (G1); k tile T+1 is first read in phase
//             2T+2, by G0 after barrier 4T+3.
// What it reaches, the elimination runs and the per-phase stamps: profiles/r06_pp_lab.txt, DESIGN.md 4 / 8 / 9.
#pragma once
#include "gemm_epilogue.h"


namespace fern {

typedef short pp_bf16x8 __attribute__((ext_vector_type(8)));
typedef int pp_i32x4 __attribute__((ext_vector_type(4)));
typedef int pp_i32x8 __attribute__((ext_vector_type(8)));

constexpr int PP_BM = 256, PP_BN = 256, PP_RB = 64;
constexpr int PP_UNIT = 256 * PP_RB;        // 16 KiB: the A (or W) rows of one k tile
constexpr int PP_SLOT = 2 * PP_UNIT;        // 32 KiB
constexpr int PP_SCALE_SLOTS = 4, PP_SCALE_BYTES = 2048;      // block-scaled family: one dword per (row, 128 k) for 512 rows

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until all but the `younger` most recent vector-memory operations of this wave are done (wave-uniform value, 0 .. 9+)
__device__ __forceinline__ void pp_wait_vmcnt_dyn(int younger) {
    if (younger == 8) { pp_wait_vmcnt<8>(); return; }      // the steady states of the two operand kinds first (one compare each)
    if (younger == 9) { pp_wait_vmcnt<9>(); return; }
    switch (younger) {
        case 0: pp_wait_vmcnt<0>(); break;
        case 1: pp_wait_vmcnt<1>(); break;
        case 2: pp_wait_vmcnt<2>(); break;
        case 3: pp_wait_vmcnt<3>(); break;
        case 4: pp_wait_vmcnt<4>(); break;
        case 5: pp_wait_vmcnt<5>(); break;
        case 6: pp_wait_vmcnt<6>(); break;
        case 7: pp_wait_vmcnt<7>(); break;
        case 8: pp_wait_vmcnt<8>(); break;
        case 9: pp_wait_vmcnt<9>(); break;
        case 10: pp_wait_vmcnt<10>(); break;
        case 11: pp_wait_vmcnt<11>(); break;
        default: pp_wait_vmcnt<12>(); break;
    }
}


// Quantising epilogue (gemm_mx8_kernel's): bias (+ GELU) in the accumulator layout, each 32x32 tile turned through a private LDS patch so
// that a lane holds 16 consecutive columns of ONE row (2 lanes per row = one 32-column MX block): block maximum -> E8M0 byte -> 16 e4m3fn
// bytes, one 16-byte store per lane.  The caller has passed the barrier after which no wave reads the operand ring any more.
template <int TM, int TN>
__device__ __forceinline__ void pp_quant_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], char* smem, int row_w, int col_w, int wave, int lane, int l31, int lh) {
    constexpr int PS = 36;
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PS);
    const int rr = lane >> 1, hh = lane & 1;
    unsigned char* C8 = reinterpret_cast<unsigned char*>(p.C);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row0 = row_w + i * 32, col0 = col_w + j * 32;
            if (row0 >= p.M || col0 >= p.N) continue;
            const float bia = p.bias ? p.bias[col0 + l31] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v2 = {acc[i][j][r] + bia, acc[i][j][r + 1] + bia};
                if (p.epi == EPI_BIAS_GELU) v2 = gelu_tanh2(v2);
                patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * PS + l31] = v2[0];
                patch[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh) * PS + l31] = v2[1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 v4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v4[q] = *reinterpret_cast<const f32x4*>(patch + rr * PS + hh * 16 + q * 4);
            float am = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(v4[q][e]));
            am = fmaxf(am, __shfl_xor(am, 1));
            const unsigned e8 = mx_scale_byte(am);
            const float inv = mx_inv_scale(e8);
            uint4 o;
            o.x = pack4_fp8(v4[0][0] * inv, v4[0][1] * inv, v4[0][2] * inv, v4[0][3] * inv);
            o.y = pack4_fp8(v4[1][0] * inv, v4[1][1] * inv, v4[1][2] * inv, v4[1][3] * inv);
            o.z = pack4_fp8(v4[2][0] * inv, v4[2][1] * inv, v4[2][2] * inv, v4[2][3] * inv);
            o.w = pack4_fp8(v4[3][0] * inv, v4[3][1] * inv, v4[3][2] * inv, v4[3][3] * inv);
            const int row = row0 + rr;
            if (row < p.M) {
                *reinterpret_cast<uint4*>(C8 + (long)row * p.ldc + col0 + hh * 16) = o;
                if (hh == 0) p.mxc[mx_scale_offset(row, col0 >> 5, p.mxc_rows)] = (unsigned char)e8;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
}

// MX = false: bf16 operands (p.Ab / p.Wb bf16 bit patterns, strides in elements).  MX = true: e4m3fn bytes + E8M0 block scales
// (p.mxa / p.mxw, mx_scale_offset layout), optionally the quantising epilogue (p.out_mx8).
template <bool MX, int NB, int VAR = 0>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmParams p) {
    constexpr int BM = PP_BM, BN = PP_BN, WM = 128, WN = 64, TM = 4, TN = 2, WAVES_N = 4;
    constexpr int RB = PP_RB, UNIT = PP_UNIT, SLOT = PP_SLOT;
    constexpr int ES = MX ? 1 : 2;
    constexpr int RING = NB * SLOT;
    constexpr int GPU_ = 2;                              // LDS-DMA instructions per wave and unit (16 pieces of 1 KiB over 8 waves)

    __shared__ __attribute__((aligned(1024))) char smem[RING + (MX ? PP_SCALE_SLOTS * PP_SCALE_BYTES : 0)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // waves w and w + 4 share a SIMD: one of each group per SIMD
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware bijective workgroup -> tile map (n fastest inside an XCD's contiguous run)
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int bm = swz / nbn, bn = swz % nbn;

    // ---- staging sources: a unit is 16 pieces of 1 KiB = 16 rows x 64 B each; wave w stages pieces w and w + 8 of every unit.
    // LDS-DMA writes lane-linear (lane l -> byte 16 l of the piece), so the bank swizzle is applied on the SOURCE chunk:
    // chunk c of row r is stored at position c ^ ((r >> 2) & 3).
    const char* srcA[GPU_];
    const char* srcW[GPU_];
    {
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int j = 0; j < GPU_; ++j) {
            const int r = (wave + 8 * j) * 16 + (lane >> 2);
            int row = bm * BM + r;
            row = row < p.M ? row : p.M - 1;
            srcA[j] = reinterpret_cast<const char*>(p.Ab) + (long)row * p.lda * ES + chunk * 16;
            row = bn * BN + r;
            row = row < p.N ? row : p.N - 1;
            srcW[j] = reinterpret_cast<const char*>(p.Wb) + (long)row * p.ldw * ES + chunk * 16;
        }
    }
    // block scales: 512 rows x 1 dword per 128 k = 8 pieces of 256 B; wave w stages piece w (A rows 64 w .. for w < 4, W rows behind)
    const char* srcS = nullptr;
    long stepS = 0;
    if (MX) {
        const int r = wave * 64 + lane;
        if (r < BM) {
            int row = bm * BM + r;
            row = row < p.M ? row : p.M - 1;
            srcS = reinterpret_cast<const char*>(p.mxa) + (long)row * 4;
            stepS = p.mxa_rows * 4;
        } else {
            int row = bn * BN + (r - BM);
            row = row < p.N ? row : p.N - 1;
            srcS = reinterpret_cast<const char*>(p.mxw) + (long)row * 4;
            stepS = p.mxw_rows * 4;
        }
    }
#ifdef FERN_GEMM_TRACE      // timing experiment (dbg bit 4 = 16): pieces of 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B, same bytes
    if (p.packed & 16) {
#pragma unroll
        for (int j = 0; j < GPU_; ++j) {
            const int r = (wave + 8 * j) * 8 + (lane >> 3);
            srcA[j] = reinterpret_cast<const char*>(p.Ab) + (long)((bm * BM + r) % p.M) * p.lda * ES + (lane & 7) * 16;
            srcW[j] = reinterpret_cast<const char*>(p.Wb) + (long)((bn * BN + r) % p.N) * p.ldw * ES + (lane & 7) * 16;
        }
    }
#endif
    auto stage_unit = [&](const char* const (&src)[GPU_], int unit, int T) {
        const int slot = T % NB;
#ifdef FERN_GEMM_TRACE
        if (p.packed & 16) {      // unit T covers rows 128 (T & 1) .. of the tile and the 128-byte k range T >> 1
#pragma unroll
            for (int j = 0; j < GPU_; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long)(T >> 1) * 128 + (long)(T & 1) * 128 * (unit ? p.ldw : p.lda) * ES),
                                                 (__attribute__((address_space(3))) void*)(smem + slot * SLOT + unit * UNIT + (wave + 8 * j) * 1024), 16, 0, 0);
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < GPU_; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long)T * RB),
                                             (__attribute__((address_space(3))) void*)(smem + slot * SLOT + unit * UNIT + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    auto stage_one = [&](const char* const (&src)[GPU_], int unit, int T, int j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long)T * RB),
                                         (__attribute__((address_space(3))) void*)(smem + (T % NB) * SLOT + unit * UNIT + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    auto stage_scales = [&](int T) {      // T even: the dwords of 128-k tile T / 2
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcS + (long)(T >> 1) * stepS),
                                         (__attribute__((address_space(3))) void*)(smem + RING + ((T >> 1) % PP_SCALE_SLOTS) * PP_SCALE_BYTES + wave * 256), 4, 0, 0);
    };
    // vector-memory operations a wave issues for k tile T (W unit first, then -- MX, T even -- the scale piece, then the A unit)
    auto ops_of = [&](int T) { return 2 * GPU_ + ((MX && !(T & 1)) ? 1 : 0); };

#ifdef FERN_GEMM_TRACE      // tools/probe/pp_lab.hip only: timing experiments (results are then garbage) and (-DPP_STAMPS) per-wave cycle stamps
    const int dbg = p.packed;      // bit 0: no staging after the prologue; bit 1: no LDS reads in the loop; bit 2: no MFMAs; bit 3: no vmcnt waits in the loop
    long long* tr = p.trace ? p.trace + ((long)blockIdx.x * 8 + wave) * FERN_GEMM_TRACE_SLOTS : nullptr;
    int tslot = 0;
#ifdef PP_STAMPS
#define PP_STAMP()                                                                                              \
    do {                                                                                                        \
        if (tr && tslot < FERN_GEMM_TRACE_SLOTS) {                                                              \
            const long long c_ = (long long)__builtin_readcyclecounter();                                       \
            if (lane == 0) tr[tslot] = c_;                                                                      \
            ++tslot;                                                                                            \
        }                                                                                                       \
    } while (0)
#else
#define PP_STAMP() do {} while (0)
#endif
#define PP_DBG(bit) (dbg & (bit))
#else
#define PP_STAMP() do {} while (0)
#define PP_DBG(bit) 0
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- fragment addresses: lane (l31, lh) reads, of its row, the 16-byte chunks lh and 2 + lh (bf16: the k groups 8 lh .. +7 of
    // the tile's two 16-k MFMA steps; fp8: k = 16 lh .. +15 and 32 + 16 lh .. +15 of the one 64-k step)
    const int sw = (l31 >> 2) & 3;
    const int c0 = (lh ^ sw) * 16, c1 = ((2 + lh) ^ sw) * 16;
    const int a_row = (wr * WM + l31) * RB;                       // + i * 32 * RB per accumulator row block
    const int b_row = UNIT + (wc * WN + l31) * RB;                // + n * 32 * RB

    pp_i32x4 af[2][2], bf[TN][2];                                   // [row block of the phase][chunk]
    int sa[2] = {0, 0}, sb[TN] = {0, 0};
    auto read_b = [&](int slot, int T) {
        const char* base = smem + slot * SLOT + b_row;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            bf[n][0] = *reinterpret_cast<const pp_i32x4*>(base + n * 32 * RB + c0);
            bf[n][1] = *reinterpret_cast<const pp_i32x4*>(base + n * 32 * RB + c1);
        }
        if (MX) {
            const unsigned* Ss = reinterpret_cast<const unsigned*>(smem + RING + ((T >> 1) % PP_SCALE_SLOTS) * PP_SCALE_BYTES);
            const int sh = 16 * (T & 1) + 8 * lh;
#pragma unroll
            for (int n = 0; n < TN; ++n) sb[n] = (int)(Ss[BM + wc * WN + n * 32 + l31] >> sh);
        }
    };
    auto read_a = [&](int slot, int T, int half) {
        const char* base = smem + slot * SLOT + a_row + half * 64 * RB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i][0] = *reinterpret_cast<const pp_i32x4*>(base + i * 32 * RB + c0);
            af[i][1] = *reinterpret_cast<const pp_i32x4*>(base + i * 32 * RB + c1);
        }
        if (MX) {
            const unsigned* Ss = reinterpret_cast<const unsigned*>(smem + RING + ((T >> 1) % PP_SCALE_SLOTS) * PP_SCALE_BYTES);
            const int sh = 16 * (T & 1) + 8 * lh;
#pragma unroll
            for (int i = 0; i < 2; ++i) sa[i] = (int)(Ss[wr * WM + half * 64 + i * 32 + l31] >> sh);
        }
    };
    // `mid(q)` (q = 0, 1) is called after the first and after the second quarter of the phase's MFMAs: VAR 1 issues the phase's two
    // LDS-DMA instructions there, in the shadow of the matrix pipe (an MFMA holds the vector issue port for 8 of its 32 / 64 cycles)
    auto mfmas = [&](int half, auto&& mid) {
        __builtin_amdgcn_s_setprio(1);
        if (MX) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const pp_i32x8 a8 = __builtin_shufflevector(af[i][0], af[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const pp_i32x8 b8 = __builtin_shufflevector(bf[n][0], bf[n][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[2 * half + i][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[2 * half + i][n], 0, 0, 0, sa[i], 0, sb[n]);
                    if (i == 0) { __builtin_amdgcn_sched_barrier(0); mid(n); __builtin_amdgcn_sched_barrier(0); }
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[2 * half + i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pp_bf16x8, af[i][kk]), __builtin_bit_cast(pp_bf16x8, bf[n][kk]),
                                                                                      acc[2 * half + i][n], 0, 0, 0);
                    if (kk == 0) { __builtin_amdgcn_sched_barrier(0); mid(i); __builtin_amdgcn_sched_barrier(0); }
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: k tiles 0 .. NB-2 go out at once (96 KiB of the 128 KiB ring), then k tile 0 is awaited
    const int nk = p.K / (RB / ES);
    const int npro = nk < NB - 1 ? nk : NB - 1;
    int younger = 0;
    for (int T = 0; T < npro; ++T) {
        stage_unit(srcW, 1, T);
        if (MX && !(T & 1)) stage_scales(T);
        stage_unit(srcA, 0, T);
        if (T > 0) younger += ops_of(T);
    }
    pp_wait_vmcnt_dyn(younger);
    barrier();
    if (wr == 1) barrier();                               // group 1 runs half a phase behind group 0 from here on

    int slot = 0;
    PP_STAMP();
    for (int T = 0; T < nk; ++T) {
        const int Ts = T + NB - 1;                        // the k tile staged during k tile T (into the slot k tile T - 1 has left)
        const bool st = Ts < nk && !PP_DBG(1);
        // phase 2T: W fragments + upper A rows; stage the W unit
        if (!PP_DBG(2)) {
            read_b(slot, T);
            read_a(slot, T, 0);
        }
        if (st && VAR == 0) {
            stage_unit(srcW, 1, Ts);
            if (MX && !(Ts & 1)) stage_scales(Ts);
        }
        barrier();
#ifdef PP_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_STAMP();
#endif
        if (!PP_DBG(4))
            mfmas(0, [&](int q) {
                if (VAR == 1 && st) {
                    stage_one(srcW, 1, Ts, q);
                    if (MX && q == 1 && !(Ts & 1)) stage_scales(Ts);
                }
            });
        PP_STAMP();
        barrier();
        PP_STAMP();
        // phase 2T+1: lower A rows; stage the A unit; k tile T+1 must have landed before the next phase reads it
        if (!PP_DBG(2)) read_a(slot, T, 1);
        if (st && VAR == 0) stage_unit(srcA, 0, Ts);
        if (T + 1 < nk && !PP_DBG(8)) {
            // the operations this wave has issued after the last one of k tile T+1: k tiles T+2 .. (VAR 1: the A unit of the newest k
            // tile goes out in THIS phase's MFMA section, i.e. after this wait)
            const int last = Ts < nk ? Ts : nk - 1;       // newest k tile this wave has issued anything of
            int y = 0;
            for (int X = T + 2; X <= last; ++X) y += ops_of(X);
            if (VAR == 1 && st) y -= GPU_;
            pp_wait_vmcnt_dyn(PP_DBG(1) ? 0 : y);
        }
        barrier();
#ifdef PP_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_STAMP();
#endif
        if (!PP_DBG(4))
            mfmas(1, [&](int q) {
                if (VAR == 1 && st) stage_one(srcA, 0, Ts, q);
            });
        PP_STAMP();
        barrier();
        PP_STAMP();
        slot = slot + 1 == NB ? 0 : slot + 1;
    }
    if (wr == 0) barrier();                               // every wave has executed the same number of barriers

    if (!MX || !p.out_mx8) {
        gemm_epilogue<BM, BN, WM, WN, TM, TN, WAVES_N, true, MX ? 2 : 1, !MX>(p, acc, bm, bn, nbn, wr, wc, l31, lh, tid);
        return;
    }
    if constexpr (MX) pp_quant_epilogue<TM, TN>(p, acc, smem, bm * BM + wr * WM, bn * BN + wc * WN, wave, lane, l31, lh);
}


}  // namespace fern
