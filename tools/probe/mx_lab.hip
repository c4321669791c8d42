// Lab bench of the block-scaled fp8 GEMM: a persistent-workgroup form of gemm_bf16.hip's gemm_mx8_kernel that
// stages the next tile's first k steps under the current tile's epilogue, and a quantising epilogue that needs no LDS turn
// (operands swapped in the MFMA, so a lane holds 16 columns of ONE row; the two lane halves trade dwords with v_permlane32_swap).
// Stand-alone (does not link the library): random e4m3 bytes + E8M0 scales, a naive device reference for correctness, hipEvent
// timing and per-workgroup phase stamps (100 MHz realtime counter).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLAB_Q: pipelined variants 2, 3 only | -DLAB_ALL: + variants 4-9 (sweep 1 of profiles/r03_mx_lab.txt
//   numbered them 1-7)] [-DLAB_FASTGELU: the family's GELU before round 3] tools/probe/mx_lab.hip -o tools/probe/mx_lab
//   tools/probe/mx_lab M N K variant epi quant grid     (epi 0 bias, 1 bias+GELU, 3 bias+residual; quant 1 = fp8 + scales out)
#include "../../fashionern_aaai2024_amd/csrc/gemm_epilogue.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace fern {
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TRACE_SLOTS = 128;

#ifdef LAB_FASTGELU      // the family's GELU before round 3 (A&S 7.1.28)
#define LAB_GELU gelu_fast2
#else
#define LAB_GELU gelu_tanh2
#endif

// QT: accumulators hold the TRANSPOSED 32x32 tile (lane = output row, registers = 16 columns) -- quantising epilogue only
template <int BM, int BN, int WM, int WN, int RB, int STAGES, int MINW, bool QT>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, MINW) void mx8p_kernel(GemmParams p, long long* trace, int dbg) {
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int C4 = RB / 16;
    constexpr int RPP = 64 / C4;
    constexpr int FSH = RB == 64 ? 2 : 1;
    constexpr int FMASK = C4 - 1;
    constexpr int PIECES = ROWS / RPP;
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    constexpr int PPW = PIECES / NW;
    constexpr int SP = ROWS / 64;
    constexpr int SPW = (SP + NW - 1) / NW;
    constexpr int DATA = ROWS * RB;
    constexpr int TILE = (DATA + ROWS * 4 + 1023) / 1024 * 1024;
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");

    __shared__ __attribute__((aligned(1024))) char smem[STAGES * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    const int G = gridDim.x;                               // G % 8 == 0 or G == nwg
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcount = xcd < r8 ? q8 + 1 : q8;              // tiles of this XCD's run
    const int nk = p.K / RB;

    const char* src[PPW];
    const char* ssrc[SPW];
    long sstep[SPW];
    auto set_tile = [&](int bm, int bn) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = wave + NW * j;
            const int trow = piece * RPP + lane / C4;
            const int chunk = (lane & FMASK) ^ ((trow >> FSH) & FMASK);
            if (trow < BM) {
                int row = bm * BM + trow;
                row = row < p.M ? row : p.M - 1;
                src[j] = reinterpret_cast<const char*>(p.Ab) + (long)row * p.lda + chunk * 16;
            } else {
                int row = bn * BN + (trow - BM);
                row = row < p.N ? row : p.N - 1;
                src[j] = reinterpret_cast<const char*>(p.Wb) + (long)row * p.ldw + chunk * 16;
            }
        }
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int sp = (wave + NW * j) % SP;
            const int trow = sp * 64 + lane;
            if (trow < BM) {
                int row = bm * BM + trow;
                row = row < p.M ? row : p.M - 1;
                ssrc[j] = reinterpret_cast<const char*>(p.mxa) + (long)row * 4;
                sstep[j] = p.mxa_rows * 4;
            } else {
                int row = bn * BN + (trow - BM);
                row = row < p.N ? row : p.N - 1;
                ssrc[j] = reinterpret_cast<const char*>(p.mxw) + (long)row * 4;
                sstep[j] = p.mxw_rows * 4;
            }
        }
    };
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = wave + NW * j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long)kt * RB),
                                             (__attribute__((address_space(3))) void*)(smem + buf * TILE + piece * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int sp = (wave + NW * j) % SP;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc[j] + (RB == 64 ? kt >> 1 : kt) * sstep[j]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * TILE + DATA + sp * 256), 4, 0, 0);
        }
    };
    auto prologue = [&]() {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) stage(t, t);
    };

    f32x16 acc[TM][TN];
    const int sw = (l31 >> FSH) & FMASK;
    auto compute = [&](int buf, int kt) {
        const char* As = smem + buf * TILE;
        const char* Ws = As + BM * RB;
        const unsigned* Ss = reinterpret_cast<const unsigned*>(As + DATA);
        int sa[TM], sb[TN];
        const int sh = RB == 64 ? 16 * (kt & 1) + 8 * lh : 8 * lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) sa[i] = (int)(Ss[wm * WM + i * 32 + l31] >> sh);
#pragma unroll
        for (int j = 0; j < TN; ++j) sb[j] = (int)(Ss[BM + wn * WN + j * 32 + l31] >> sh);
#pragma unroll
        for (int e = 0; e < RB / 64; ++e) {
            const int pc0 = ((4 * e + lh) ^ sw) * 16, pc1 = ((4 * e + 2 + lh) ^ sw) * 16;
            i32x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char* r = As + (wm * WM + i * 32 + l31) * RB;
                af[i] = __builtin_shufflevector(*reinterpret_cast<const i32x4*>(r + pc0), *reinterpret_cast<const i32x4*>(r + pc1), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const char* r = Ws + (wn * WN + j * 32 + l31) * RB;
                bf[j] = __builtin_shufflevector(*reinterpret_cast<const i32x4*>(r + pc0), *reinterpret_cast<const i32x4*>(r + pc1), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (QT) {
                        if (e == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[j], af[i], acc[i][j], 0, 0, 0, sb[j], 0, sa[i]);
                        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[j], af[i], acc[i][j], 0, 0, 2, sb[j], 2, sa[i]);
                    } else {
                        if (e == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, sa[i], 0, sb[j]);
                        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 2, sa[i], 2, sb[j]);
                    }
                }
        }
    };

    long long* tr = trace ? trace + ((long)blockIdx.x * NW + wave) * TRACE_SLOTS : nullptr;
    int ts = 0;
    auto stamp = [&]() {
        if (tr && lane == 0 && ts + 1 < TRACE_SLOTS) { tr[ts] = (long long)wall_clock64(); tr[ts + 1] = (long long)__builtin_readcyclecounter(); }
        ts += 2;
    };

    int local = blockIdx.x >> 3;                            // index within this XCD's run; next = local + G / 8
    if (local >= xcount) return;
    {
        const int swz = xbase + local;
        set_tile(swz / nbn, swz % nbn);
    }
    stamp();
    prologue();
    while (true) {
        const int swz = xbase + local;
        const int bm = swz / nbn, bn = swz % nbn;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        int slot = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int younger = nk - 1 - kt < STAGES - 2 ? nk - 1 - kt : STAGES - 2;
            if (STAGES >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PPW + SPW)) : "memory");
            else if (STAGES >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + SPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt == 0) stamp();
            if (kt + STAGES - 1 < nk && !(dbg & 1)) {
                int fill = slot + STAGES - 1;
                fill = fill >= STAGES ? fill - STAGES : fill;
                stage(fill, kt + STAGES - 1);
            }
            compute((dbg & 2) ? 0 : slot, kt);
            slot = slot + 1 == STAGES ? 0 : slot + 1;
        }
        stamp();
        const int row_w = bm * BM + wm * WM, col_w = bn * BN + wn * WN;
        float bia[QT ? TN : 1][16];
        if (QT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = col_w + j * 32 + 8 * q + 4 * lh;
                    const f32x4 b4 = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    bia[j][4 * q] = b4[0]; bia[j][4 * q + 1] = b4[1]; bia[j][4 * q + 2] = b4[2]; bia[j][4 * q + 3] = b4[3];
                }
        }
        const int nlocal = local + (G >> 3);
        const bool more = G != nwg && nlocal < xcount;
        if (more) {
            __builtin_amdgcn_s_barrier();                   // every wave is done reading the ring
            const int nswz = xbase + nlocal;
            set_tile(nswz / nbn, nswz % nbn);
            prologue();
        }
        if (dbg & 4) {
            if (acc[0][0][0] == 123.456f) p.C[tid] = acc[TM - 1][TN - 1][3];
        } else if (!QT) {
            gemm_epilogue<BM, BN, WM, WN, TM, TN, WAVES_N, true>(p, acc, bm, bn, nbn, wm, wn, l31, lh, tid);
        } else {
            unsigned char* C8 = reinterpret_cast<unsigned char*>(p.C);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = row_w + i * 32 + l31;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col0 = col_w + j * 32;
                    if (row_w + i * 32 >= p.M || col0 >= p.N) continue;
                    float v[16];
                    float am = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 v2 = {acc[i][j][r] + bia[j][r], acc[i][j][r + 1] + bia[j][r + 1]};
                        if (p.epi == EPI_BIAS_GELU) v2 = LAB_GELU(v2);
                        v[r] = v2[0]; v[r + 1] = v2[1];
                        am = fmaxf(am, fmaxf(fabsf(v2[0]), fabsf(v2[1])));
                    }
                    {
                        const u32x2 sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(am), __float_as_uint(am), false, false);
                        am = fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
                    }
                    const unsigned e8 = mx_scale_byte(am);
                    const float inv = mx_inv_scale(e8);
                    unsigned d0 = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
                    unsigned d1 = pack4_fp8(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
                    unsigned d2 = pack4_fp8(v[8] * inv, v[9] * inv, v[10] * inv, v[11] * inv);
                    unsigned d3 = pack4_fp8(v[12] * inv, v[13] * inv, v[14] * inv, v[15] * inv);
                    // lane half 0 keeps columns 0-15, half 1 columns 16-31: upper half of the first operand <-> lower half of the second
                    const u32x2 s02 = __builtin_amdgcn_permlane32_swap(d0, d2, false, false);
                    const u32x2 s13 = __builtin_amdgcn_permlane32_swap(d1, d3, false, false);
                    d0 = s02[0]; d2 = s02[1]; d1 = s13[0]; d3 = s13[1];
                    if (row < p.M) {
                        uint4 o;
                        o.x = d0; o.y = d2; o.z = d1; o.w = d3;
                        *reinterpret_cast<uint4*>(C8 + (long)row * p.ldc + col0 + lh * 16) = o;
                        if (lh == 0) p.mxc[mx_scale_offset(row, col0 >> 5, p.mxc_rows)] = (unsigned char)e8;
                    }
                }
            }
        }
        stamp();
        if (!more) break;
        local = nlocal;
    }
}


// ---- software-pipelined form ------------------------------------------------------------------------------------------------------
// 64-byte tile rows (one 64-k MFMA step per ring slot).  The fragments of step kt+1 are read from LDS WHILE the MFMAs of step kt
// issue (A fragments reloaded in place right behind the MFMA group that used them, W fragments double-buffered), so a wave's MFMA
// stream does not stop for LDS latency and the per-step barrier falls between two groups of already-loaded operands.
// Ring: at step kt the slot of tile kt is free (its fragments are in registers) and is refilled with tile kt + STAGES.
template <int BM, int BN, int WM, int WN, int STAGES, int MINW, bool QT>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, MINW) void mx8q_kernel(GemmParams p, long long* trace, int dbg) {
    constexpr int RB = 64;
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int RPP = 16;                               // tile rows per 1 KiB piece
    constexpr int PIECES = ROWS / RPP;
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    static_assert(BM % RPP == 0, "a piece is all A or all W");
    constexpr int PPW = PIECES / NW;
    constexpr int SP = ROWS / 64;
    constexpr int SPW = (SP + NW - 1) / NW;
    constexpr int DATA = ROWS * RB;
    constexpr int TILE = (DATA + ROWS * 4 + 1023) / 1024 * 1024;
    constexpr int NOPS = PPW + SPW;                       // vm operations per wave and tile
    static_assert(NOPS >= TM, "at least one staging operation per MFMA row");
    static_assert(STAGES >= 2 && STAGES <= 5, "ring depth");

    __shared__ __attribute__((aligned(1024))) char smem[STAGES * TILE + 256];      // + the dummy copies' landing pad

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    const int G = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcount = xcd < r8 ? q8 + 1 : q8;
    const int nk = p.K / RB;                              // even (K % 128 == 0)

    // per-lane 32-bit byte offsets from the (wave-uniform) operand bases: operands < 4 GiB
    unsigned src[PPW], ssrc[SPW];
    const char* const Abase = reinterpret_cast<const char*>(p.Ab);
    const char* const Wbase = reinterpret_cast<const char*>(p.Wb);
    const char* const SAbase = reinterpret_cast<const char*>(p.mxa);
    const char* const SWbase = reinterpret_cast<const char*>(p.mxw);
    auto set_tile = [&](int bm, int bn) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = wave + NW * j;
            const int trow = piece * RPP + (lane >> 2);
            const int chunk = (lane & 3) ^ ((trow >> 2) & 3);
            if (piece * RPP < BM) {
                int row = bm * BM + trow;
                row = row < p.M ? row : p.M - 1;
                src[j] = (unsigned)row * (unsigned)p.lda + chunk * 16;
            } else {
                int row = bn * BN + (trow - BM);
                row = row < p.N ? row : p.N - 1;
                src[j] = (unsigned)row * (unsigned)p.ldw + chunk * 16;
            }
        }
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int sp = (wave + NW * j) % SP;
            const int trow = sp * 64 + lane;
            if (sp * 64 < BM) {
                int row = bm * BM + trow;
                row = row < p.M ? row : p.M - 1;
                ssrc[j] = (unsigned)row * 4;
            } else {
                int row = bn * BN + (trow - BM);
                row = row < p.N ? row : p.N - 1;
                ssrc[j] = (unsigned)row * 4;
            }
        }
    };
    // vm operation `op` (0 .. NOPS-1) of tile kt into slot buf: PPW 1 KiB row pieces, then SPW 256-byte scale pieces
    auto stage_op = [&](int op, int buf, int kt) {
        if (op < PPW) {
            const int piece = wave + NW * op;
            const char* base = (piece * RPP < BM ? Abase : Wbase) + (long)kt * RB;          // wave-uniform
#ifdef LAB_SADDR
            // scalar base + 32-bit lane offset form of the same instruction (the builtin selects the 64-bit vector address form)
            const unsigned ldsa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + buf * TILE + piece * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(ldsa), "v"(src[op]), "s"(base) : "memory");
#else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + src[op]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * TILE + piece * 1024), 16, 0, 0);
#endif
        } else {
            const int j = op - PPW;
            const int sp = (wave + NW * j) % SP;
            const char* base = sp * 64 < BM ? SAbase + (long)(kt >> 1) * p.mxa_rows * 4 : SWbase + (long)(kt >> 1) * p.mxw_rows * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ssrc[j]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * TILE + DATA + sp * 256), 4, 0, 0);
        }
    };
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int op = 0; op < NOPS; ++op) stage_op(op, buf, kt);
    };
    // Past the last tile the ring is topped up with NOPS 4-byte copies of scale bytes into the dead slot's scale area, so that the wait
    // count of the k loop is the same immediate at every step.
    auto dummy_op = [&]() {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SAbase + ssrc[0]),
                                         (__attribute__((address_space(3))) void*)(smem + STAGES * TILE), 4, 0, 0);
    };
    // vmcnt is an immediate: wait until at most `groups` tile groups of this wave are still in flight
#define WAIT_GROUPS(g) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((g) * NOPS) : "memory")
    f32x16 acc[TM][TN];
    i32x8 af[TM], bf[2][TN];
    int sa[TM], sb[2][TN];
    const int sw = (l31 >> 2) & 3;
    const int pc0 = (lh ^ sw) * 16, pc1 = ((2 + lh) ^ sw) * 16;
    const int arow = (wm * WM + l31) * RB, wrow = (BM + wn * WN + l31) * RB;     // byte offsets of this lane's first A / W row in a slot
    const int asc = (wm * WM + l31) * 4 + lh, wsc = (BM + wn * WN + l31) * 4 + lh;     // + lh: the lane half's scale byte of a 64-k step
    auto read_a = [&](int i, int slot, int kt) {
        const char* r = smem + slot * TILE + arow + i * 32 * RB;
        af[i] = __builtin_shufflevector(*reinterpret_cast<const i32x4*>(r + pc0), *reinterpret_cast<const i32x4*>(r + pc1), 0, 1, 2, 3, 4, 5, 6, 7);
        sa[i] = *reinterpret_cast<const unsigned char*>(smem + slot * TILE + DATA + asc + i * 128 + 2 * (kt & 1));
    };
    auto read_b = [&](int j, int slot, int kt, int par) {
        const char* r = smem + slot * TILE + wrow + j * 32 * RB;
        bf[par][j] = __builtin_shufflevector(*reinterpret_cast<const i32x4*>(r + pc0), *reinterpret_cast<const i32x4*>(r + pc1), 0, 1, 2, 3, 4, 5, 6, 7);
        sb[par][j] = *reinterpret_cast<const unsigned char*>(smem + slot * TILE + DATA + wsc + j * 128 + 2 * (kt & 1));
    };
    auto mfma_row = [&](int i, int par) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (QT) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[par][j], af[i], acc[i][j], 0, 0, 0, sb[par][j], 0, sa[i]);
            else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[par][j], acc[i][j], 0, 0, 0, sa[i], 0, sb[par][j]);
        }
        // MFMAs are pure values to the optimiser, which otherwise moves a whole step's group across the barrier to the next step's
        // (leaving one step all reads, one all MFMAs); an empty asm that "updates" the accumulators pins them in program order
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]) : : "memory");
    };

    long long* tr = trace ? trace + ((long)blockIdx.x * NW + wave) * TRACE_SLOTS : nullptr;
    int ts = 0;
    auto stamp = [&]() {
        if (tr && lane == 0 && ts + 1 < TRACE_SLOTS) { tr[ts] = (long long)wall_clock64(); tr[ts + 1] = (long long)__builtin_readcyclecounter(); }
        ts += 2;
    };

    int local = blockIdx.x >> 3;
    if (local >= xcount) return;
    {
        const int swz = xbase + local;
        set_tile(swz / nbn, swz % nbn);
    }
    stamp();
#pragma unroll
    for (int t = 0; t < STAGES; ++t) stage(t, t);
    while (true) {
        const int swz = xbase + local;
        const int bm = swz / nbn, bn = swz % nbn;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        // tile 0 -> registers (nk >= STAGES + 2: the ring is full)
        WAIT_GROUPS(STAGES - 1);
        __builtin_amdgcn_s_barrier();
        stamp();
#pragma unroll
        for (int i = 0; i < TM; ++i) read_a(i, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_b(j, 0, 0, 0);
        int slot = 0;                                       // slot of tile kt
        // One k step: PAR = kt & 1 picks the W fragment buffer; NEXT: tile kt + 1 exists; FILL: tile kt + STAGES exists;
        // YOUNGER: tile groups issued after tile kt + 1 at this point.  All four are compile-time at the call sites.
        // One k step, the same code for every kt (a statically unrolled tail lets the compiler sink the tail's MFMAs below its
        // reads): the wait count and the refill are the only runtime (scalar) choices; the last step reads a stale slot ahead.
        auto step = [&](int kt, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            const int nslot = slot + 1 == STAGES ? 0 : slot + 1;
            WAIT_GROUPS(STAGES - 2);                           // tile kt + 1 landed: STAGES - 2 groups (real or dummy) were issued after it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool fill = kt + STAGES < nk;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                mfma_row(i, PAR);
                // this row's share of the refill of slot kt (LDS-DMA issue costs the wave ~100 cycles a piece: spread under the MFMAs)
#pragma unroll
                for (int op = i * NOPS / TM; op < (i + 1) * NOPS / TM; ++op) {
                    if (fill) stage_op(op, slot, kt + STAGES);
                    else dummy_op();
                }
                read_a(i, nslot, kt + 1);
                if (i < TN) read_b(i, nslot, kt + 1, PAR ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (TN > TM) {
#pragma unroll
                for (int j = TM; j < TN; ++j) read_b(j, nslot, kt + 1, PAR ^ 1);
            }
            slot = nslot;
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, std::integral_constant<int, 0>{});
            step(kt + 1, std::integral_constant<int, 1>{});
        }
        stamp();
        const int row_w = bm * BM + wm * WM, col_w = bn * BN + wn * WN;
        float bia[QT ? TN : 1][16];
        if (QT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = col_w + j * 32 + 8 * q + 4 * lh;
                    const f32x4 b4 = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    bia[j][4 * q] = b4[0]; bia[j][4 * q + 1] = b4[1]; bia[j][4 * q + 2] = b4[2]; bia[j][4 * q + 3] = b4[3];
                }
        }
        const int nlocal = local + (G >> 3);
        const bool more = G != nwg && nlocal < xcount;
        if (more) {
            __builtin_amdgcn_s_barrier();                   // every wave's fragments of the last tile are in registers
            const int nswz = xbase + nlocal;
            set_tile(nswz / nbn, nswz % nbn);
#pragma unroll
            for (int t = 0; t < STAGES; ++t) stage(t, t);
        }
        if (dbg & 4) {
            if (acc[0][0][0] == 123.456f) p.C[tid] = acc[TM - 1][TN - 1][3];
        } else if (!QT) {
            if (p.epi == EPI_BIAS_RESIDUAL) plain_epilogue<EPI_BIAS_RESIDUAL, false, false>(p, acc, row_w, col_w, l31, lh);
            else if (p.epi == EPI_BIAS_GELU) plain_epilogue<EPI_BIAS_GELU, true, false, true>(p, acc, row_w, col_w, l31, lh);
            else plain_epilogue<EPI_BIAS, true, false>(p, acc, row_w, col_w, l31, lh);
        } else {
            unsigned char* C8 = reinterpret_cast<unsigned char*>(p.C);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = row_w + i * 32 + l31;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col0 = col_w + j * 32;
                    if (row_w + i * 32 >= p.M || col0 >= p.N) continue;
                    float v[16];
                    float am = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 v2 = {acc[i][j][r] + bia[j][r], acc[i][j][r + 1] + bia[j][r + 1]};
                        if (p.epi == EPI_BIAS_GELU) v2 = LAB_GELU(v2);
                        v[r] = v2[0]; v[r + 1] = v2[1];
                        am = fmaxf(am, fmaxf(fabsf(v2[0]), fabsf(v2[1])));
                    }
                    {
                        const u32x2 sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(am), __float_as_uint(am), false, false);
                        am = fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
                    }
                    const unsigned e8 = mx_scale_byte(am);
                    const float inv = mx_inv_scale(e8);
                    unsigned d0 = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
                    unsigned d1 = pack4_fp8(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
                    unsigned d2 = pack4_fp8(v[8] * inv, v[9] * inv, v[10] * inv, v[11] * inv);
                    unsigned d3 = pack4_fp8(v[12] * inv, v[13] * inv, v[14] * inv, v[15] * inv);
                    const u32x2 s02 = __builtin_amdgcn_permlane32_swap(d0, d2, false, false);
                    const u32x2 s13 = __builtin_amdgcn_permlane32_swap(d1, d3, false, false);
                    d0 = s02[0]; d2 = s02[1]; d1 = s13[0]; d3 = s13[1];
                    if (row < p.M) {
                        uint4 o;
                        o.x = d0; o.y = d2; o.z = d1; o.w = d3;
                        *reinterpret_cast<uint4*>(C8 + (long)row * p.ldc + col0 + lh * 16) = o;
                        if (lh == 0) p.mxc[mx_scale_offset(row, col0 >> 5, p.mxc_rows)] = (unsigned char)e8;
                    }
                }
            }
        }
        stamp();
        if (!more) break;
        local = nlocal;
    }
}

// naive reference: one thread per output element
__global__ void mx_ref_kernel(GemmParams p, float* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)p.M * p.N) return;
    const int m = idx / p.N, n = idx % p.N;
    const unsigned char* a = reinterpret_cast<const unsigned char*>(p.Ab) + (long)m * p.lda;
    const unsigned char* w = reinterpret_cast<const unsigned char*>(p.Wb) + (long)n * p.ldw;
    double tot = 0;
    for (int b = 0; b < p.K / 32; ++b) {
        float s = 0;
        for (int k = 0; k < 32; ++k) s += __builtin_amdgcn_cvt_f32_fp8(a[b * 32 + k], 0) * __builtin_amdgcn_cvt_f32_fp8(w[b * 32 + k], 0);
        const int ea = p.mxa[mx_scale_offset(m, b, p.mxa_rows)], ew = p.mxw[mx_scale_offset(n, b, p.mxw_rows)];
        tot += (double)s * exp2((double)(ea - 127)) * exp2((double)(ew - 127));
    }
    float v = (float)tot + (p.bias ? p.bias[n] : 0.f);
    if (p.epi == EPI_BIAS_GELU) { f32x2 g = LAB_GELU(f32x2{v, v}); v = g[0]; }
    if (p.epi == EPI_BIAS_RESIDUAL) v += p.R[(long)m * p.N + n];
    out[idx] = v;
}
}  // namespace fern

using namespace fern;

struct Variant { const char* name; int bm, bn, threads; void (*k)(GemmParams, long long*, int); void (*kq)(GemmParams, long long*, int); int per_cu; };
#define V(BM, BN, WM, WN, RB, ST, MW, PC) {#BM "x" #BN " w" #WM "x" #WN " rb" #RB " st" #ST, BM, BN, (BM / WM) * (BN / WN) * 64, \
    mx8p_kernel<BM, BN, WM, WN, RB, ST, MW, false>, mx8p_kernel<BM, BN, WM, WN, RB, ST, MW, true>, PC}
#define VQ_PLACEHOLDER
static const Variant kV[] = {
#ifndef LAB_Q
    V(128, 128, 64, 64, 128, 2, 2, 2),     // 0: the library's config 0 shape
    V(256, 256, 128, 64, 64, 4, 1, 1),     // 1: config 10 shape (8 waves of 128x64)
#else
    {}, {},
#endif
#define VQ(BM, BN, WM, WN, ST, MW, PC) {#BM "x" #BN " w" #WM "x" #WN " pipelined st" #ST, BM, BN, (BM / WM) * (BN / WN) * 64, \
    mx8q_kernel<BM, BN, WM, WN, ST, MW, false>, mx8q_kernel<BM, BN, WM, WN, ST, MW, true>, PC}
    VQ(256, 256, 128, 64, 4, 1, 1),        // 2: pipelined, 8 waves of 128x64
    VQ(256, 128, 128, 64, 3, 2, 2),        // 3: pipelined, 4 waves of 128x64, two workgroups per CU
#ifdef LAB_ALL
    V(256, 128, 64, 64, 64, 3, 2, 2),      // 4: config 9 shape
    V(256, 256, 64, 64, 128, 2, 1, 1),     // 5: config 7 shape (16 waves)
    V(256, 128, 128, 64, 64, 3, 2, 2),     // 6: 4 waves of 128x64, 2 per CU
    V(256, 256, 128, 64, 64, 3, 1, 1),     // 7
    V(128, 256, 64, 64, 64, 3, 2, 2),      // 8: 8 waves of 64x64, wide
    V(256, 256, 128, 64, 128, 2, 1, 1),    // 9
#endif
};

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12608, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
    const int vi = argc > 4 ? atoi(argv[4]) : 2, epi = argc > 5 ? atoi(argv[5]) : 1, quant = argc > 6 ? atoi(argv[6]) : 1;
    int G = argc > 7 ? atoi(argv[7]) : 0;                 // 0: one workgroup per tile (not persistent)
    const int dbg = argc > 8 ? atoi(argv[8]) : 0;         // timing experiments (results garbage): 1 no staging after the prologue, 2 every step reads slot 0, 4 no epilogue
    const Variant& v = kV[vi];
    const long nwg = (long)((M + v.bm - 1) / v.bm) * ((N + v.bn - 1) / v.bn);
    if (G <= 0 || G >= nwg) G = (int)nwg;
    else G = (G + 7) / 8 * 8;
    unsigned char *A, *W, *sa, *sw, *C, *sc;
    float *bias, *R, *ref;
    const long sab = (long)(K / 128) * M * 4, swb = (long)(K / 128) * N * 4;
    hipMalloc(&A, (size_t)M * K); hipMalloc(&W, (size_t)N * K); hipMalloc(&sa, sab); hipMalloc(&sw, swb);
    hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&sc, (size_t)M * N / 32 + 64); hipMalloc(&bias, N * 4); hipMalloc(&R, (size_t)M * N * 4);
    hipMalloc(&ref, (size_t)M * N * 4);
    srand(3);
    {
        std::vector<unsigned char> h((size_t)std::max(M, N) * K), hs(std::max(sab, swb));
        for (auto& b : h) { b = rand() & 0xFF; if ((b & 0x7F) == 0x7F) b ^= 1; if ((b & 0x78) == 0x78) b &= ~0x40; }
        for (auto& b : hs) b = 118 + rand() % 6;
        hipMemcpy(A, h.data(), (size_t)M * K, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data() + 1, (size_t)N * K - 1, hipMemcpyHostToDevice);
        hipMemcpy(sa, hs.data(), sab, hipMemcpyHostToDevice);
        hipMemcpy(sw, hs.data(), swb, hipMemcpyHostToDevice);
        std::vector<float> hb(N), hr((size_t)M * N);
        for (auto& f : hb) f = (float)rand() / RAND_MAX - 0.5f;
        for (auto& f : hr) f = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
        hipMemcpy(R, hr.data(), (size_t)M * N * 4, hipMemcpyHostToDevice);
    }
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A); p.Wb = reinterpret_cast<const unsigned short*>(W);
    p.C = reinterpret_cast<float*>(C); p.bias = bias; p.R = R;
    p.lda = K; p.ldw = K; p.ldc = N; p.M = M; p.N = N; p.K = K; p.epi = epi; p.fp8 = 2;
    p.mxa = sa; p.mxw = sw; p.mxa_rows = M; p.mxw_rows = N;
    p.out_mx8 = quant; p.mxc = sc; p.mxc_rows = M;
    p.out_bf16 = (!quant && epi != EPI_BIAS_RESIDUAL) ? 1 : 0;
    hipStream_t s;
    hipStreamCreate(&s);
    auto kern = quant ? v.kq : v.k;
    auto launch = [&](long long* trace) { hipLaunchKernelGGL(kern, dim3(G), dim3(v.threads), 0, s, p, trace, dbg); };
    const int nw = v.threads / 64;
    long long* trace;
    hipMalloc(&trace, (size_t)G * nw * TRACE_SLOTS * 8);
    hipMemset(trace, 0, (size_t)G * nw * TRACE_SLOTS * 8);
    // correctness
    hipMemset(C, 0, (size_t)M * N * 4);
    launch(nullptr);
    if (hipStreamSynchronize(s) != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipLaunchKernelGGL(mx_ref_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, s, p, ref);
    hipStreamSynchronize(s);
    {
        std::vector<float> hr((size_t)M * N);
        hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, rms = 0;
        long bad = 0;
        if (quant) {
            std::vector<unsigned char> hc((size_t)M * N), hs((size_t)M * N / 32);
            hipMemcpy(hc.data(), C, hc.size(), hipMemcpyDeviceToHost);
            hipMemcpy(hs.data(), sc, hs.size(), hipMemcpyDeviceToHost);
            for (long m = 0; m < M; ++m)
                for (long n = 0; n < N; ++n) {
                    const int e = hs[mx_scale_offset(m, n >> 5, M)];
                    const unsigned char b = hc[m * N + n];
                    const int ex = (b >> 3) & 15, mant = b & 7;
                    float f = ex ? ldexpf(1.f + mant / 8.f, ex - 7) : ldexpf(mant / 8.f, -6);
                    if (b & 0x80) f = -f;
                    const float got = ldexpf(f, e - 127), want = hr[m * N + n];
                    // block maximum sets the step: half a step of the block's scale is 2^(e-127) * 2^(ex-7-4); allow one full step
                    const float tol = ldexpf(1.f, e - 127) * ldexpf(1.f, (ex ? ex : 1) - 7 - 3) * 1.01f + 1e-6f;
                    if (std::fabs(got - want) > tol) { if (bad < 5) printf("bad (%ld,%ld): got %g want %g (e %d byte %02x)\n", m, n, got, want, e, b); ++bad; }
                    rms += (double)want * want;
                }
            printf("quantised output: %ld of %ld elements off by more than one fp8 step (rms %.3f)\n", bad, (long)M * N, std::sqrt(rms / ((double)M * N)));
        } else if (p.out_bf16) {
            std::vector<unsigned short> hc((size_t)M * N);
            hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < hc.size(); ++i) {
                unsigned u = (unsigned)hc[i] << 16;
                float f; memcpy(&f, &u, 4);
                worst = std::max(worst, (double)std::fabs(f - hr[i]) / (std::fabs(hr[i]) + 1.0));
            }
            printf("bf16 output: max relative error %.3e\n", worst);
        } else {
            std::vector<float> hc((size_t)M * N);
            hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < hc.size(); ++i) worst = std::max(worst, (double)std::fabs(hc[i] - hr[i]) / (std::fabs(hr[i]) + 1.0));
            printf("fp32 output: max relative error %.3e\n", worst);
        }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = (int)(6e12 / (2.0 * M * N * K)) + 20;
    for (int i = 0; i < warm; ++i) launch(nullptr);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) launch(nullptr);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s grid %d (%ld tiles) epi %d quant %d: %.2f us per launch = %.0f TFLOP/s\n", v.name, G, nwg, epi, quant, ms * 50, 2.0 * M * N * K / (ms * 5e-5) / 1e12);
    // phase stamps of one launch (wave 0 of every workgroup): per tile {first data ready, loop end, epilogue end}
    for (int i = 0; i < 5; ++i) launch(nullptr);
    launch(trace);
    hipStreamSynchronize(s);
    std::vector<long long> ht((size_t)G * nw * TRACE_SLOTS);
    hipMemcpy(ht.data(), trace, ht.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = ht[0], t1 = 0;
    for (int g = 0; g < G; ++g) t0 = std::min(t0, ht[(size_t)g * nw * TRACE_SLOTS]);
    double wait = 0, loop = 0, epil = 0, first_wait = 0, loop_cyc = 0;
    long tiles = 0;
    for (int g = 0; g < G; ++g) {
        const long long* t = &ht[(size_t)g * nw * TRACE_SLOTS];
        long long prev = t[0];
        for (int i = 0; 2 * (1 + 3 * i + 2) + 1 < TRACE_SLOTS && t[2 * (1 + 3 * i + 2)]; ++i) {
            const long long ready = t[2 * (1 + 3 * i)], lend = t[2 * (2 + 3 * i)], eend = t[2 * (3 + 3 * i)];
            if (i == 0) first_wait += ready - prev; else wait += ready - prev;
            loop += lend - ready; epil += eend - lend;
            loop_cyc += t[2 * (2 + 3 * i) + 1] - t[2 * (1 + 3 * i) + 1];
            prev = eend; t1 = std::max(t1, eend); ++tiles;
        }
    }
    printf("stamps: kernel span %.2f us; per tile: first-tile data wait %.2f us (x%d), later-tile wait %.2f us, k loop %.2f us (%.0f cycles: %.2f GHz), epilogue %.2f us; %ld tiles\n",
           (t1 - t0) * 0.01, first_wait * 0.01 / G, G, tiles > G ? wait * 0.01 / (tiles - G) : 0.0, loop * 0.01 / tiles, loop_cyc / tiles, loop_cyc / (loop * 10.0), epil * 0.01 / tiles, tiles);
    return 0;
}
