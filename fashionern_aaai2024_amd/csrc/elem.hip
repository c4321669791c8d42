// Row-wise kernels of the encode -> fuse -> rank path (HBM-bound; one wave64 per row, 16-byte accesses).
//
// LayerNorm (CLIP eps 1e-5, BERT eps 1e-12 with fused residual), F.normalize / VisualSR.l2norm,
// patch means, embedding gathers, and the scalar-gate / attention-pooling tails of
// CombinerSimple (fusion_model.py:92-94) and VisualSR (fusion_model.py:149-154).
#include "kernels.h"

namespace fern {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves per workgroup, one row each
constexpr int MAXV = 5;             // row width <= 64 lanes * 4 floats * MAXV = 1280 (2 x 640 for the CLIP4Cir Combiner)

// Wave-wide reductions on the DPP network (quad swaps, half-row and row mirrors: every lane then holds its 16-lane row's value) plus
// four readlanes added in a fixed order -- ~10 issue slots, against six dependent ds_bpermute round trips (~100 cycles each) for the
// __shfl_xor butterfly.  The row kernels are one wave per row with two or three reductions each: that latency, not the bytes, was
// most of a LayerNorm launch.  Results are the same for every lane and depend only on the row (batch-invariant).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }   // (readlane is an int builtin)
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);      // row_half_mirror
    v += dpp_move<0x140>(v);      // row_mirror
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
// maximum over the aligned group of 4 / 8 consecutive lanes (the lanes that share a 32-element MX block)
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    return fmaxf(v, dpp_move<0x4E>(v));
}
__device__ __forceinline__ float oct_max(float v) { v = quad_max(v); return fmaxf(v, dpp_move<0x141>(v)); }

// A row of width d (d % 4 == 0, d <= 1024) held as up to MAXV float4 per lane: element c = (i*64 + lane)*4.
struct RowRegs {
    f32x4 v[MAXV];
};

__device__ __forceinline__ void row_load(RowRegs& r, const float* x, int d, int lane) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c < d) t = *reinterpret_cast<const f32x4*>(x + c);
        r.v[i] = t;
    }
}
__device__ __forceinline__ void row_store(const RowRegs& r, float* y, int d, int lane) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) *reinterpret_cast<f32x4*>(y + c) = r.v[i];
    }
}
__device__ __forceinline__ float row_sumsq(const RowRegs& r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += r.v[i][0] * r.v[i][0] + r.v[i][1] * r.v[i][1] + r.v[i][2] * r.v[i][2] + r.v[i][3] * r.v[i][3];
    return wave_sum(s);
}

// y = (x - mean) / sqrt(var + eps) * gamma + beta with the two-pass variance torch uses
__device__ __forceinline__ void row_layernorm(RowRegs& r, const float* gamma, const float* beta, int d, int lane, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += r.v[i][0] + r.v[i][1] + r.v[i][2] + r.v[i][3];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = r.v[i][e] - mean; q += t * t; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) r.v[i][e] = (r.v[i][e] - mean) * rstd * g[e] + bb[e];
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* res, const float* gamma, const float* beta,
                                                        float* y, long rows, int d, long ldx, long ldy, float eps) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    RowRegs r;
    row_load(r, x + row * ldx, d, lane);
    if (res) {
        RowRegs q;
        row_load(q, res + row * ldx, d, lane);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) r.v[i] += q.v[i];
    }
    row_layernorm(r, gamma, beta, d, lane, eps);
    row_store(r, y + row * ldy, d, lane);
}

__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const float* x, const float* gamma, const float* beta, unsigned short* y,
                                                             long rows, int d, long ldx, long ldy, float eps) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    RowRegs r;
    row_load(r, x + row * ldx, d, lane);
    row_layernorm(r, gamma, beta, d, lane, eps);
    unsigned short* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            ushort4 o;
            o.x = f32_to_bf16_bits(r.v[i][0]); o.y = f32_to_bf16_bits(r.v[i][1]);
            o.z = f32_to_bf16_bits(r.v[i][2]); o.w = f32_to_bf16_bits(r.v[i][3]);
            *reinterpret_cast<ushort4*>(yr + c) = o;
        }
    }
}

// LayerNorm fused with per-row fp8 (e4m3fn) quantisation: scale = max|y| / 448, y8 = fp8(y / scale)
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* x, const float* gamma, const float* beta, unsigned char* y,
                                                            float* scale, long rows, int d, long ldx, long ldy, float eps) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    RowRegs r;
    row_load(r, x + row * ldx, d, lane);
    row_layernorm(r, gamma, beta, d, lane, eps);
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) am = fmaxf(am, fmaxf(fmaxf(fabsf(r.v[i][0]), fabsf(r.v[i][1])), fmaxf(fabsf(r.v[i][2]), fabsf(r.v[i][3]))));
    }
    am = wave_max(am);
    const float sc = am > 0.f ? am * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    unsigned char* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) *reinterpret_cast<unsigned*>(yr + c) = pack4_fp8(r.v[i][0] * inv, r.v[i][1] * inv, r.v[i][2] * inv, r.v[i][3] * inv);
    }
}

// Per-row fp8 quantisation of a bf16 or fp32 matrix (one wave per row, rows up to 8 * 512 = 4096 wide, d % 8 == 0):
// scale = max|x| / 448, y8 = fp8(x / scale).  Used for the attention / GELU outputs (bf16) and the weight matrices (fp32).
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const unsigned short* xb, const float* xf, long ldx, unsigned char* y, long ldy,
                                                                float* scale, long rows, int d) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[8][8];
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (c < d) {
            if (xb) {
                const uint4 t = *reinterpret_cast<const uint4*>(xb + row * ldx + c);
                const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[i][2 * e] = __uint_as_float(w[e] << 16); v[i][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xf + row * ldx + c), b = *reinterpret_cast<const f32x4*>(xf + row * ldx + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[i][e]));
        }
    }
    am = wave_max(am);
    const float sc = am > 0.f ? am * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < d) {
            uint2 o;
            o.x = pack4_fp8(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
            o.y = pack4_fp8(v[i][4] * inv, v[i][5] * inv, v[i][6] * inv, v[i][7] * inv);
            *reinterpret_cast<uint2*>(y + row * ldy + c) = o;
        }
    }
}

// ---- MX (block-scaled) e4m3fn quantisers: one E8M0 byte per 32 consecutive k, the operand form of
// v_mfma_scale_f32_32x32x64_f8f6f4 (gemm_bf16.hip: gemm_mx8_kernel).  e = the smallest power of two that brings the block's
// maximum to <= 448, read off the maximum's exponent / mantissa bits (448 = 1.75 * 2^8); the scaling x * 2^(127-e) is exact.

// LayerNorm fused with the MX quantiser (a lane holds 4 consecutive elements: a 32-block is 8 lanes).  XB: the rows are bf16 (the
// block-scaled mode's residual stream), 8-byte loads of the same 4 elements per lane.
__device__ __forceinline__ void row_load_bf16(RowRegs& r, const unsigned short* x, int d, int lane) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            const uint2 w = *reinterpret_cast<const uint2*>(x + c);
            t[0] = __uint_as_float(w.x << 16); t[1] = __uint_as_float(w.x & 0xffff0000u);
            t[2] = __uint_as_float(w.y << 16); t[3] = __uint_as_float(w.y & 0xffff0000u);
        }
        r.v[i] = t;
    }
}
template <bool XB>
__global__ __launch_bounds__(256) void layernorm_mx8_kernel(const float* x, const unsigned short* xb, const float* gamma, const float* beta, unsigned char* y,
                                                            unsigned char* scales, long srows, long rows, int d, long ldx, long ldy, float eps) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    RowRegs r;
    if (XB) row_load_bf16(r, xb + row * ldx, d, lane);
    else row_load(r, x + row * ldx, d, lane);
    row_layernorm(r, gamma, beta, d, lane, eps);
    unsigned char* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        float am = fmaxf(fmaxf(fabsf(r.v[i][0]), fabsf(r.v[i][1])), fmaxf(fabsf(r.v[i][2]), fabsf(r.v[i][3])));
        am = oct_max(am);
        if (c < d) {
            const unsigned e = mx_scale_byte(am);
            const float inv = mx_inv_scale(e);
            *reinterpret_cast<unsigned*>(yr + c) = pack4_fp8(r.v[i][0] * inv, r.v[i][1] * inv, r.v[i][2] * inv, r.v[i][3] * inv);
            if ((lane & 7) == 0) scales[mx_scale_offset(row, c >> 5, srows)] = (unsigned char)e;
        }
    }
}

// ---- several rows per wave (round 4) ------------------------------------------------------------------------------------------
// One wave per row is one memory round trip, two dependent wave reductions and one store per row, back to back: at 12608 rows x
// 768 columns the LayerNorm launches ran at 2.8 TB/s (bf16 rows in, fp8 out) -- latency, not bytes.  Here a wave owns RPW consecutive
// rows: all their loads are issued before the first reduction (one round trip per RPW rows), the 2 x RPW reductions interleave, and
// gamma / beta are fetched once per wave.  The width is a template parameter (d = 256 NV: no per-chunk bounds tests).  The per-row
// arithmetic is the one-row kernels' statement for statement, and the kernel is used for EVERY row count of a width it covers, so a
// row's bits still do not depend on the batch it travels in.
// IN: 0 fp32 rows, 1 bf16 rows.  OUT: 0 fp32 (may alias the input), 1 e4m3fn + E8M0 block scales (the MX quantiser above).
template <int NV, int RPW, int IN, int OUT>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const void* xin, const float* gamma, const float* beta, void* yout, unsigned char* scales,
                                                             long srows, long rows, long ldx, long ldy, float eps) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    f32x4 v[RPW][NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < rows ? row0 + r : rows - 1;      // a short last group re-reads the last row (not stored)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (IN == 1) {
                const uint2 w = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(xin) + row * ldx + c);
                v[r][i] = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
            } else {
                v[r][i] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(xin) + row * ldx + c);
            }
        }
    }
    f32x4 g[NV], bb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + (i * 64 + lane) * 4);
        bb[i] = *reinterpret_cast<const f32x4*>(beta + (i * 64 + lane) * 4);
    }
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
        mean[r] = wave_sum(s) / (float)D;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = v[r][i][e] - mean[r]; q += t * t; }
        rstd[r] = rsqrtf(wave_sum(q) / (float)D + eps);
    }
    unsigned e8s[RPW][NV];               // OUT == 1: the E8M0 byte of this lane's 32-block, per (row, chunk)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mean[r]) * rstd[r] * g[i][e] + bb[i][e];
            if (OUT == 0) {
                if (row < rows) *reinterpret_cast<f32x4*>(static_cast<float*>(yout) + row * ldy + c) = o;
            } else {
                float am = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
                am = oct_max(am);
                const unsigned e8 = mx_scale_byte(am);
                const float inv = mx_inv_scale(e8);
                e8s[r][i] = e8;
                if (row < rows)
                    *reinterpret_cast<unsigned*>(static_cast<unsigned char*>(yout) + row * ldy + c) = pack4_fp8(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
            }
        }
    }
    if (OUT == 1) {
        // Scales: the layout holds one dword per (128-k tile kt, row) = the bytes of blocks 4kt .. 4kt+3, i.e. of lanes 0, 8, 16, 24 (+ 32
        // for the odd kt) of chunk kt / 2 -- gathered with v_readlane into ONE dword per (row, kt), and the RPW = 4 consecutive rows of the
        // wave are 16 contiguous bytes per kt: one 16-byte store instead of 16 single-byte stores from 16 lanes of 4 instructions (a
        // byte store is a write transaction of its own; 24 per row were as many transactions as all the data stores of the row).
        static_assert(RPW == 4, "the scale gather writes the four rows of a wave as one uint4");
        const bool whole = row0 + RPW <= rows && (srows & 3) == 0;      // row0 is a multiple of 4: the 16-byte stores are aligned
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned dw[RPW];
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int e = (int)e8s[r][i];
                    dw[r] = (unsigned)__builtin_amdgcn_readlane(e, 32 * half) | ((unsigned)__builtin_amdgcn_readlane(e, 32 * half + 8) << 8) |
                            ((unsigned)__builtin_amdgcn_readlane(e, 32 * half + 16) << 16) | ((unsigned)__builtin_amdgcn_readlane(e, 32 * half + 24) << 24);
                }
                const long kt = 2 * i + half;
                unsigned* dst = reinterpret_cast<unsigned*>(scales + (kt * srows + row0) * 4);
                if (lane == 0) {
                    if (whole) *reinterpret_cast<uint4*>(dst) = uint4{dw[0], dw[1], dw[2], dw[3]};
                    else
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
                            if (row0 + r < rows) dst[r] = dw[r];
                }
            }
    }
}
constexpr int LN_RPW = 4;
// widths the several-rows kernel covers (256, 512, 768, 1024): launch it and return true
template <int IN, int OUT>
static bool launch_ln_rows(const void* x, const float* gamma, const float* beta, void* y, unsigned char* scales, long srows, long rows, int d,
                           long ldx, long ldy, float eps, hipStream_t s) {
    if (d % 256 || d > 1024) return false;
    // gamma / beta are read as f32x4 and (OUT = 1) four rows' scale dwords leave as one 16-byte store: the public entry points accept
    // any 4-byte-aligned view, so a pointer that is not 16-byte aligned takes the one-row kernel instead
    if (((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || (OUT == 1 && ((uintptr_t)scales & 15))) return false;
    const dim3 grid((unsigned)((rows + ROWS_PER_BLOCK * LN_RPW - 1) / (ROWS_PER_BLOCK * LN_RPW)));
    switch (d / 256) {
        case 1: hipLaunchKernelGGL((layernorm_rows_kernel<1, LN_RPW, IN, OUT>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
        case 2: hipLaunchKernelGGL((layernorm_rows_kernel<2, LN_RPW, IN, OUT>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
        case 3: hipLaunchKernelGGL((layernorm_rows_kernel<3, LN_RPW, IN, OUT>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
        default: hipLaunchKernelGGL((layernorm_rows_kernel<4, LN_RPW, IN, OUT>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
    }
    return true;
}

// ---- bf16 rows in, MX fp8 out: half a wave per row (round 4, tools/probe/ln_lab.hip) ---------------------------------------------
// The block-scaled mode's LayerNorm reads the bf16 residual stream the GEMM before it has just written (from other XCDs: nothing of
// it is in this CU's L2) and is a single round of waves: its time is the latency of that round, not its 29 MB.  Measured in the chain
// writer -> LayerNorm -> reader at 12608 x 768: the four-rows-per-wave kernel above 15.1 us, one row per wave 10.7, this form 9.1
// (a byte-for-byte copy of the same shape: 8.7).  A lane holds 8 consecutive elements of each 256-column chunk of ITS half-wave's
// row -- one 16-byte load and one 8-byte store per chunk, 512 contiguous bytes per half-wave -- and a 32-element MX block is 4 lanes.
// The fp8 bytes leave with nt stores: the next kernel reads them from other XCDs anyway.  Statistics: the row's 32 lanes on the DPP
// network, the two 16-lane rows of the half added in a fixed order; used for EVERY row count of a width it covers (batch-invariant).
// F32IN (round 6, FERN_PREC_MX8_MLP / _MX8_IMG: block-scaled operands over the FP32 residual stream): the same kernel on fp32 rows --
// two 16-byte loads per chunk and lane -- instead of the four-rows-per-wave kernel (14.0 us at 12608 x 768, 22 launches per step).
template <int NV, bool F32IN = false>
__global__ __launch_bounds__(256) void layernorm_half_kernel(const void* xin, const float* gamma, const float* beta, unsigned char* y, unsigned char* scales,
                                                             long srows, long rows, long ldx, long ldy, float eps) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const long rowp = ((long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * 2;
    if (rowp >= rows) return;
    const long row = rowp + lh;
    const long rrow = row < rows ? row : rows - 1;             // an odd row count: the last wave's upper half re-reads the last row (not stored)
    float v[NV][8];
    if (F32IN) {
        const float* x = reinterpret_cast<const float*>(xin);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + rrow * ldx + (i * 32 + l31) * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x + rrow * ldx + (i * 32 + l31) * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; }
        }
    } else {
        const unsigned short* x = reinterpret_cast<const unsigned short*>(xin);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint4 w = *reinterpret_cast<const uint4*>(x + rrow * ldx + (i * 32 + l31) * 8);
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][2 * e] = __uint_as_float(ww[e] << 16); v[i][2 * e + 1] = __uint_as_float(ww[e] & 0xffff0000u); }
        }
    }
    float g[NV][8], bb[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 32 + l31) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[i][e] = g0[e]; g[i][4 + e] = g1[e]; bb[i][e] = b0[e]; bb[i][4 + e] = b1[e]; }
    }
    auto half_sum = [&](float t) {      // over the 32 lanes of this lane's half
        t += dpp_move<0xB1>(t);
        t += dpp_move<0x4E>(t);
        t += dpp_move<0x141>(t);
        t += dpp_move<0x140>(t);
        const float lo = lane_f(t, 0) + lane_f(t, 16), hi = lane_f(t, 32) + lane_f(t, 48);
        return lh ? hi : lo;
    };
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v[i][e];
    const float mean = half_sum(sum) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = v[i][e] - mean; q += t * t; }
    const float rstd = rsqrtf(half_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float o[8];
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = (v[i][e] - mean) * rstd * g[i][e] + bb[i][e]; am = fmaxf(am, fabsf(o[e])); }
        am = quad_max(am);
        const unsigned e8 = mx_scale_byte(am);
        const float inv = mx_inv_scale(e8);
        if (row < rows) {
            unsigned* dst = reinterpret_cast<unsigned*>(y + row * ldy + (i * 32 + l31) * 8);
            __builtin_nontemporal_store(pack4_fp8(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv), dst);
            __builtin_nontemporal_store(pack4_fp8(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv), dst + 1);
            // blocks 8i .. 8i+7 of the row, 4 lanes each (mx_scale_offset: k tile = block / 4, byte = block % 4)
            if ((l31 & 3) == 0) scales[mx_scale_offset(row, i * 8 + (l31 >> 2), srows)] = (unsigned char)e8;
        }
    }
}
template <bool F32IN>
static bool launch_ln_half(const void* x, const float* gamma, const float* beta, unsigned char* y, unsigned char* scales, long srows, long rows, int d,
                           long ldx, long ldy, float eps, hipStream_t s) {
    if (d % 256 || d > 1024 || (ldx & 7) || (ldy & 7) || ((uintptr_t)x & 15) || ((uintptr_t)y & 7) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return false;
    const dim3 grid((unsigned)((rows + ROWS_PER_BLOCK * 2 - 1) / (ROWS_PER_BLOCK * 2)));
    switch (d / 256) {
        case 1: hipLaunchKernelGGL((layernorm_half_kernel<1, F32IN>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
        case 2: hipLaunchKernelGGL((layernorm_half_kernel<2, F32IN>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
        case 3: hipLaunchKernelGGL((layernorm_half_kernel<3, F32IN>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
        default: hipLaunchKernelGGL((layernorm_half_kernel<4, F32IN>), grid, dim3(256), 0, s, x, gamma, beta, y, scales, srows, rows, ldx, ldy, eps); break;
    }
    return true;
}

// Patch rows of a [b, 3, img, img] image batch, block-scale quantised: row (image, gy, gx) = the 3 x patch x patch pixels of one
// patch in (channel, y, x) order -- the k order of conv1's [width, 3 * patch * patch] weight -- so that the patch embedding of the
// block-scaled mode is a plain MX GEMM.  One wave per patch; a lane holds 4 consecutive x of one (channel, y) per step.
__global__ __launch_bounds__(256) void im2col_mx8_kernel(const float* images, unsigned char* y, unsigned char* scales, long srows, long rows,
                                                         int img, int patch, int grid) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int pp = patch * patch, d = 3 * pp, g2 = grid * grid;
    const long b = row / g2;
    const int gy = (int)(row % g2) / grid, gx = (int)(row % g2) % grid;
    const float* src = images + b * 3L * img * img + (long)(gy * patch) * img + gx * patch;
    unsigned char* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int k = (i * 64 + lane) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < d) {
            const int ch = k / pp, rem = k % pp;
            v = *reinterpret_cast<const f32x4*>(src + ((long)ch * img + rem / patch) * img + rem % patch);
        }
        const float am = oct_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        if (k < d) {
            const unsigned e = mx_scale_byte(am);
            const float inv = mx_inv_scale(e);
            *reinterpret_cast<unsigned*>(yr + k) = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
            if ((lane & 7) == 0) scales[mx_scale_offset(row, k >> 5, srows)] = (unsigned char)e;
        }
    }
}

// The same patch rows as bf16 (round to nearest even): the patch embedding of the bf16-operand modes (FERN_PREC_BF16 / _MX8_MLP /
// _MX8_IMG, round 6) is a plain bf16 GEMM -- the im2col-loading fp32 GEMM took 166 us of a 4.9 ms step where this pair takes ~40.
__global__ __launch_bounds__(256) void im2col_bf16_kernel(const float* images, unsigned short* y, long rows, int img, int patch, int grid) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int pp = patch * patch, d = 3 * pp, g2 = grid * grid;
    const long b = row / g2;
    const int gy = (int)(row % g2) / grid, gx = (int)(row % g2) % grid;
    const float* src = images + b * 3L * img * img + (long)(gy * patch) * img + gx * patch;
    unsigned short* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int k = (i * 64 + lane) * 4;
        if (k < d) {
            const int ch = k / pp, rem = k % pp;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((long)ch * img + rem / patch) * img + rem % patch);
            uint2 o;
            o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
            *reinterpret_cast<uint2*>(yr + k) = o;
        }
    }
}

// bf16 or fp32 rows up to 4096 wide (a lane holds 8 consecutive elements: a 32-block is 4 lanes)
__global__ __launch_bounds__(256) void quantize_mx8_kernel(const unsigned short* xb, const float* xf, long ldx, unsigned char* y, long ldy,
                                                           unsigned char* scales, long srows, long rows, int d) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i * 64 + lane) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (c < d) {
            if (xb) {
                const uint4 t = *reinterpret_cast<const uint4*>(xb + row * ldx + c);
                const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xf + row * ldx + c), b = *reinterpret_cast<const f32x4*>(xf + row * ldx + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
            }
        }
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[e]));
        am = quad_max(am);
        if (c < d) {
            const unsigned e8 = mx_scale_byte(am);
            const float inv = mx_inv_scale(e8);
            uint2 o;
            o.x = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
            o.y = pack4_fp8(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
            *reinterpret_cast<uint2*>(y + row * ldy + c) = o;
            if ((lane & 3) == 0) scales[mx_scale_offset(row, c >> 5, srows)] = (unsigned char)e8;
        }
    }
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float* x, const float* x2, long ldx, float* y, long ldy, long rows, int d,
                                                     float eps, int mode) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    RowRegs r;
    row_load(r, x + row * ldx, d, lane);
    if (x2) {                                   // normalize(x + x2): utils.element_wise_sum
        RowRegs q;
        row_load(q, x2 + row * ldx, d, lane);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) r.v[i] += q.v[i];
    }
    const float nrm = sqrtf(row_sumsq(r));
    const float den = mode == 0 ? fmaxf(nrm, eps) : nrm + eps;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) r.v[i] = r.v[i] / den;
    row_store(r, y + row * ldy, d, lane);
}

// Cross-entropy of `scale * logits[row, :]` against label = row (losses/loss.py:10-14): log-sum-exp minus the diagonal logit.
__global__ __launch_bounds__(256) void ce_diag_rows_kernel(const float* logits, long ld, int n, float scale, float* row_loss) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* x = logits + row * ld;
    float m = -INFINITY;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, x[c] * scale);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) sum += expf(x[c] * scale - m);
    sum = wave_sum(sum);
    if (lane == 0) row_loss[row] = (m + logf(sum)) - x[row] * scale;
}
// out[0] = mean(v[0..n)) in a fixed summation order (one wave)
__global__ __launch_bounds__(64) void mean_scalar_kernel(const float* v, int n, float* out) {
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += 64) s += v[c];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

// y[row] = mean_{p<P} x[row*group_stride + row_add + p]
// One wave per (row, 256-column chunk); the P rows are requested EIGHT at a time and added in order p = 0, 1, ... (the sum's order --
// and so its bits -- is that of the one-load-at-a-time loop it replaces: 64 query rows x 77 tokens took 30 us on 64 waves, each
// waiting for one row at a time).
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* x, long ldx, float* y, long ldy, long n, int P, int d,
                                                        long group_stride, long row_add) {
    const int nchunk = (d + 255) / 256;
    const long item = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long row = item / nchunk;
    const int c = (int)(item % nchunk) * 256 + lane * 4;
    if (row >= n || c >= d) return;
    const float* src = x + (row * group_stride + row_add) * ldx + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 32 <= P; p += 32) {      // round 5: 32 rows in flight (128 VGPRs) -- 77 tokens are 3 round trips instead of 10 (a round trip here is ~3 us:
        f32x4 t[32];                    // the rows were just written from other XCDs); same summation order, same bits
#pragma unroll
        for (int u = 0; u < 32; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (long)(p + u) * ldx);
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += t[u];
    }
    for (; p + 8 <= P; p += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (long)(p + u) * ldx);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; p < P; ++p) acc += *reinterpret_cast<const f32x4*>(src + (long)p * ldx);
    const float invp = 1.0f / (float)P;
    *reinterpret_cast<f32x4*>(y + row * ldy + c) = acc * invp;
}

// y[i] = x[(i / group)*group_stride + (i % group) + (idx ? idx[i] : row_add)]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* x, long ldx, float* y, long ldy, long n, int d, int group,
                                                          long group_stride, long row_add, const int* idx) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const long src = (row / group) * group_stride + (row % group) + (idx ? (long)idx[row] : row_add);
    RowRegs r;
    row_load(r, x + src * ldx, d, lane);
    row_store(r, y + row * ldy, d, lane);
}

// y[i] (fp32) = widen(x[i * row_stride]) for a bf16 source (the class rows of the block-scaled mode's bf16 residual stream): one thread per
// four columns, exact widening
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const unsigned short* x, long ldx, float* y, long ldy, long n, int d, long row_stride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int d4 = d >> 2;
    if (i >= n * d4) return;
    const long row = i / d4;
    const int c = (int)(i % d4) * 4;
    const ushort4 v = *reinterpret_cast<const ushort4*>(x + row * row_stride * ldx + c);
    float4 o;
    o.x = bf16_bits_to_f32(v.x); o.y = bf16_bits_to_f32(v.y); o.z = bf16_bits_to_f32(v.z); o.w = bf16_bits_to_f32(v.w);
    *reinterpret_cast<float4*>(y + row * ldy + c) = o;
}

__global__ __launch_bounds__(256) void bert_embed_kernel(const float* cls, const float* local, const float* seq, const float* type_emb,
                                                         const float* pos_emb, const float* gamma, const float* beta, float* X,
                                                         int B, int P, int T, int d, float eps) {
    const int S = 1 + P + T;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * S) return;
    const int b = (int)(row / S), s = (int)(row % S);
    const float* src = s == 0 ? cls : (s <= P ? local + ((long)b * P + (s - 1)) * d : seq + ((long)b * T + (s - 1 - P)) * d);
    RowRegs r, t, q;
    row_load(r, src, d, lane);
    row_load(t, type_emb + (s > P ? d : 0), d, lane);
    row_load(q, pos_emb + (long)s * d, d, lane);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) r.v[i] = (r.v[i] + t.v[i]) + q.v[i];     // HF BertEmbeddings order: inputs + type, then + position
    row_layernorm(r, gamma, beta, d, lane, eps);
    row_store(r, X + row * d, d, lane);
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* text, const float* tok_emb, const float* pos_emb, float* X,
                                                         int B, int T, int d, int vocab, int* bad_flag) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * T) return;
    const int s = (int)(row % T);
    const long tok = text[row];
    RowRegs r, q;
    if (tok < 0 || tok >= vocab) {
        // nn.Embedding raises on such an id (the reference's token_embedding lookup); a kernel cannot, so the row is poisoned
        // with NaN -- which reaches every feature of this caption -- and the host-visible flag records where it happened
        // (1 + flat position); fern_sync / the next fern_text_encode on this context report it as FERN_ERR_ARG
        if (lane == 0 && bad_flag) __hip_atomic_store(bad_flag, (int)(row < 0x7FFFFFFEL ? row + 1 : 0x7FFFFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) r.v[i] = __builtin_nanf("");
        row_store(r, X + row * d, d, lane);
        return;
    }
    row_load(r, tok_emb + tok * d, d, lane);
    row_load(q, pos_emb + (long)s * d, d, lane);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) r.v[i] += q.v[i];
    row_store(r, X + row * d, d, lane);
}

// eot[b] = first position of the maximum token id (torch.argmax semantics on the reference's EOT pooling).  One wave per caption:
// a single thread walking the 77 tokens is a chain of 77 dependent memory round trips (20 us for a kernel that moves 40 KB).
__global__ __launch_bounds__(256) void text_eot_kernel(const int64_t* text, int* eot, int B, int T) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    long long best = -0x7fffffffffffffffLL - 1;
    int pos = 0x7fffffff;
    for (int s = lane; s < T; s += 64) {      // ascending positions per lane: a lane keeps its first maximum
        const long long v = text[(long)b * T + s];
        if (v > best) { best = v; pos = s; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long ob = __shfl_xor(best, off);
        const int op = __shfl_xor(pos, off);
        if (ob > best || (ob == best && op < pos)) { best = ob; pos = op; }
    }
    if (lane == 0) eot[b] = pos;
}

__global__ __launch_bounds__(256) void vit_cls_kernel(const float* cls, const float* pos, float* X, int B, int tokens, int d) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    RowRegs r, q;
    row_load(r, cls, d, lane);
    row_load(q, pos, d, lane);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) r.v[i] += q.v[i];
    row_store(r, X + row * tokens * d, d, lane);
}

// bias + partial[0] + partial[1] + ... in that order (the reduce epilogues' per-32-column partial sums of one row): the partials are
// fetched 64 at a time, one per lane, and added on the scalar path lane by lane -- the same chain of additions, bit for bit, as the
// one-dependent-load-per-step loop it replaces (which cost a memory round trip per partial on every wave)
__device__ __forceinline__ float ordered_partial_sum(float z, const float* partial, int nb, int lane) {
    for (int j0 = 0; j0 < nb; j0 += 64) {
        const float v = j0 + lane < nb ? partial[j0 + lane] : 0.0f;
        const int m = nb - j0 < 64 ? nb - j0 : 64;
        for (int j = 0; j < m; ++j) z += lane_f(v, j);
    }
    return z;
}

__global__ __launch_bounds__(256) void combiner_finalize_kernel(const float* partial, int nb, const float* b2, const float* image,
                                                                const float* text, const float* extra, float* out, long n, int d) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float z = ordered_partial_sum(b2[0], partial + row * nb, nb, lane);           // fixed order: deterministic
    const float s = 1.0f / (1.0f + expf(-z));
    RowRegs im, tx;
    row_load(im, image + row * d, d, lane);
    row_load(tx, text + row * d, d, lane);
    if (extra) {                                // CLIP4Cir Combiner: output_layer(...) + s*text + (1-s)*image (Combiner_Model.py:63-67)
        RowRegs ex;
        row_load(ex, extra + row * d, d, lane);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) im.v[i] = (ex.v[i] + tx.v[i] * s) + im.v[i] * (1.0f - s);
    } else {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) im.v[i] = tx.v[i] * s + im.v[i] * (1.0f - s);
    }
    const float den = fmaxf(sqrtf(row_sumsq(im)), 1e-12f);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) im.v[i] = im.v[i] / den;
    row_store(im, out + row * d, d, lane);
}

__global__ __launch_bounds__(256) void sr_finalize_kernel(const float* partial, int nb, const float* bc, const float* local, float* out,
                                                          long n, int d) {
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float w[13];
    float m = -INFINITY;
#pragma unroll
    for (int p = 0; p < 13; ++p) {
        const float z = ordered_partial_sum(bc[0], partial + (row * 13 + p) * nb, nb, lane);
        w[p] = z;
        m = fmaxf(m, z);
    }
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 13; ++p) { w[p] = expf(w[p] - m); sum += w[p]; }
    RowRegs acc;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) acc.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 13; ++p) {
        RowRegs r;
        row_load(r, local + (row * 13 + p) * d, d, lane);
        const float wp = w[p] / sum;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) acc.v[i] += r.v[i] * wp;
    }
    const float den = sqrtf(row_sumsq(acc)) + 1e-8f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) acc.v[i] = acc.v[i] / den;
    row_store(acc, out + row * d, d, lane);
}

// one wave per (query, named row): a coalesced d-wide dot product
__global__ __launch_bounds__(256) void gather_scores_kernel(const float* q, const float* gallery, const int* idx, float* out, int B, int m, int d) {
    const long item = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= (long)B * m) return;
    const int gi = idx[item];
    if (gi < 0) { if (lane == 0) out[item] = -INFINITY; return; }
    RowRegs a, g;
    row_load(a, q + (item / m) * d, d, lane);
    row_load(g, gallery + (long)gi * d, d, lane);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += a.v[i][0] * g.v[i][0] + a.v[i][1] * g.v[i][1] + a.v[i][2] * g.v[i][2] + a.v[i][3] * g.v[i][3];
    s = wave_sum(s);
    if (lane == 0) out[item] = s;
}

// ---- ModifiedResNet (open_clip RN50x4 image tower) helpers -------------------------------------------------------
// Stem conv1: 3 input channels -> K = 27, far too thin for MFMA; a direct convolution (0.1 % of the tower's flops).
// One thread = one output pixel x 4 output channels; NCHW image in, NHWC activation out, BatchNorm folded, ReLU.
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* img, const float* w, const float* bias, float* out, int B, int S,
                                                        int cout) {
    const int So = S / 2, c4n = cout / 4;
    const long total = (long)B * So * So * c4n;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c4 = (int)(t % c4n);
    const long pix = t / c4n;
    const int x = (int)(pix % So), y = (int)((pix / So) % So), b = (int)(pix / ((long)So * So));
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = 2 * y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = 2 * x + kx - 1;
                float v = 0.f;
                if (yy >= 0 && yy < S && xx >= 0 && xx < S) v = img[(((long)b * 3 + c) * S + yy) * S + xx];
                const int k = (c * 3 + ky) * 3 + kx;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, w[(c4 * 4 + e) * 27 + k], acc[e]);
            }
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
    *reinterpret_cast<f32x4*>(out + pix * cout + c4 * 4) = acc;
}

// The same convolution with one thread = one output pixel x ALL output channels (C4N groups of 4): the 27 image taps are loaded once
// per pixel instead of once per channel group, the weights come from an LDS image laid out [tap][channel] (one broadcast
// ds_read_b128 per tap and channel group), and a thread stores its pixel's whole channel vector contiguously.  Each output is the
// same fma chain (bias, then taps in (c, ky, kx) order): bit-identical to the kernel above.
template <int C4N>
__global__ __launch_bounds__(256) void stem_conv_px_kernel(const float* img, const float* w, const float* bias, float* out, int B, int S) {
    constexpr int COUT = 4 * C4N;
    __shared__ f32x4 ws[27][C4N];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) {
        const int k = i / COUT, co = i % COUT;
        reinterpret_cast<float*>(&ws[0][0])[i] = w[co * 27 + k];
    }
    __syncthreads();
    const int So = S / 2;
    const long total = (long)B * So * So;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total) return;
    const int x = (int)(pix % So), y = (int)((pix / So) % So), b = (int)(pix / ((long)So * So));
    f32x4 acc[C4N];
#pragma unroll
    for (int c4 = 0; c4 < C4N; ++c4) acc[c4] = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
    float v[27];      // the pixel's taps, all requested before the first is used
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = 2 * y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = 2 * x + kx - 1;
                float t = 0.f;
                if (yy >= 0 && yy < S && xx >= 0 && xx < S) t = img[(((long)b * 3 + c) * S + yy) * S + xx];
                v[(c * 3 + ky) * 3 + kx] = t;
            }
        }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
#pragma unroll
        for (int c4 = 0; c4 < C4N; ++c4) {
            const f32x4 wv = ws[k][c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c4][e] = fmaf(v[k], wv[e], acc[c4][e]);
        }
        __builtin_amdgcn_sched_barrier(0);      // one tap's weight reads at a time (unfenced, all 270 are hoisted and spill)
    }
    float* o = out + pix * COUT;
#pragma unroll
    for (int c4 = 0; c4 < C4N; ++c4) {
        f32x4 r = acc[c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
        *reinterpret_cast<f32x4*>(o + c4 * 4) = r;
    }
}

__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float* x, float* y, int B, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k, c4n = C / 4;
    const long total = (long)B * Ho * Wo * c4n;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c4 = (int)(t % c4n);
    const long pix = t / c4n;
    const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx)
            acc += *reinterpret_cast<const f32x4*>(x + (((long)b * H + yo * k + dy) * W + xo * k + dx) * C + c4 * 4);
    acc = acc * (1.0f / (float)(k * k));
    *reinterpret_cast<f32x4*>(y + pix * C + c4 * 4) = acc;
}

// mean over the HW tokens of each image (any width C % 4 == 0): one thread per (image, 4 channels)
__global__ __launch_bounds__(256) void mean_tokens_kernel(const float* x, float* mean, int B, int HW, int C) {
    const int c4n = C / 4;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * c4n) return;
    const int c4 = (int)(t % c4n), b = (int)(t / c4n);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < HW; ++i) acc += *reinterpret_cast<const f32x4*>(x + ((long)b * HW + i) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(mean + (long)b * C + c4 * 4) = acc * (1.0f / (float)HW);
}

__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const float* x, const float* mean, const float* pos, float* T, float* T0,
                                                              int B, int HW, int C) {
    const int c4n = C / 4;
    const long total = (long)B * (HW + 1) * c4n;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c4 = (int)(t % c4n);
    const long row = t / c4n;
    const int tok = (int)(row % (HW + 1)), b = (int)(row / (HW + 1));
    const float* src = tok == 0 ? mean + (long)b * C : x + ((long)b * HW + tok - 1) * C;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c4 * 4) + *reinterpret_cast<const f32x4*>(pos + (long)tok * C + c4 * 4);
    *reinterpret_cast<f32x4*>(T + row * C + c4 * 4) = v;
    if (tok == 0) *reinterpret_cast<f32x4*>(T0 + (long)b * C + c4 * 4) = v;      // compact copy of the query (mean) token
}

static inline bool bad_width(int d) { return d <= 0 || (d & 3) || d > 64 * 4 * MAXV; }
static inline dim3 row_grid(long rows) { return dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)); }

hipError_t launch_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float* y, long rows, int d,
                            long ldx, long ldy, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (bad_width(d) || (ldx & 3) || (ldy & 3)) return hipErrorInvalidValue;
    if (!res && launch_ln_rows<0, 0>(x, gamma, beta, y, nullptr, 0, rows, d, ldx, ldy, eps, s)) return hipGetLastError();
    hipLaunchKernelGGL(layernorm_kernel, row_grid(rows), dim3(256), 0, s, x, res, gamma, beta, y, rows, d, ldx, ldy, eps);
    return hipGetLastError();
}
hipError_t launch_layernorm_bf16(const float* x, const float* gamma, const float* beta, unsigned short* y, long rows, int d, long ldx,
                                 long ldy, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (bad_width(d) || (ldx & 3) || (ldy & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_bf16_kernel, row_grid(rows), dim3(256), 0, s, x, gamma, beta, y, rows, d, ldx, ldy, eps);
    return hipGetLastError();
}
hipError_t launch_layernorm_fp8(const float* x, const float* gamma, const float* beta, unsigned char* y, float* scale, long rows, int d,
                                long ldx, long ldy, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (bad_width(d) || (ldx & 3) || (ldy & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_fp8_kernel, row_grid(rows), dim3(256), 0, s, x, gamma, beta, y, scale, rows, d, ldx, ldy, eps);
    return hipGetLastError();
}
hipError_t launch_layernorm_mx8(const float* x, const float* gamma, const float* beta, unsigned char* y, unsigned char* scales, long srows,
                                long rows, int d, long ldx, long ldy, float eps, hipStream_t s, const unsigned short* x_bf16) {
    if (rows <= 0) return hipSuccess;
    if (d <= 0 || d % 128 || d > 256 * MAXV || (ldx & 3) || (ldy & 3) || srows < rows) return hipErrorInvalidValue;
    if (x_bf16 ? launch_ln_half<false>(x_bf16, gamma, beta, y, scales, srows, rows, d, ldx, ldy, eps, s)
               : launch_ln_half<true>(x, gamma, beta, y, scales, srows, rows, d, ldx, ldy, eps, s))
        return hipGetLastError();
    if (x_bf16 ? launch_ln_rows<1, 1>(x_bf16, gamma, beta, y, scales, srows, rows, d, ldx, ldy, eps, s)
               : launch_ln_rows<0, 1>(x, gamma, beta, y, scales, srows, rows, d, ldx, ldy, eps, s))
        return hipGetLastError();
    if (x_bf16) hipLaunchKernelGGL(layernorm_mx8_kernel<true>, row_grid(rows), dim3(256), 0, s, x, x_bf16, gamma, beta, y, scales, srows, rows, d, ldx, ldy, eps);
    else hipLaunchKernelGGL(layernorm_mx8_kernel<false>, row_grid(rows), dim3(256), 0, s, x, x_bf16, gamma, beta, y, scales, srows, rows, d, ldx, ldy, eps);
    return hipGetLastError();
}
hipError_t launch_im2col_mx8(const float* images, unsigned char* y, unsigned char* scales, long srows, int b, int img, int patch, int grid,
                             hipStream_t s) {
    const long rows = (long)b * grid * grid;
    if (rows <= 0) return hipSuccess;
    const int d = 3 * patch * patch;
    if (d % 128 || d > 256 * MAXV || (patch & 3) || (img & 3) || grid * patch != img || srows < rows) return hipErrorInvalidValue;
    hipLaunchKernelGGL(im2col_mx8_kernel, row_grid(rows), dim3(256), 0, s, images, y, scales, srows, rows, img, patch, grid);
    return hipGetLastError();
}
hipError_t launch_im2col_bf16(const float* images, unsigned short* y, int b, int img, int patch, int grid, hipStream_t s) {
    const long rows = (long)b * grid * grid;
    if (rows <= 0) return hipSuccess;
    const int d = 3 * patch * patch;
    if (d % 32 || d > 256 * MAXV || (patch & 3) || (img & 3) || grid * patch != img) return hipErrorInvalidValue;
    hipLaunchKernelGGL(im2col_bf16_kernel, row_grid(rows), dim3(256), 0, s, images, y, rows, img, patch, grid);
    return hipGetLastError();
}
hipError_t launch_quantize_mx8(const unsigned short* x_bf16, const float* x_f32, long ldx, unsigned char* y, long ldy, unsigned char* scales,
                               long srows, long rows, int d, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (d <= 0 || d % 128 || d > 4096 || (ldx & 7) || (ldy & 7) || srows < rows || (x_bf16 == nullptr) == (x_f32 == nullptr)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_mx8_kernel, row_grid(rows), dim3(256), 0, s, x_bf16, x_f32, ldx, y, ldy, scales, srows, rows, d);
    return hipGetLastError();
}
hipError_t launch_quantize_rows_fp8(const unsigned short* x_bf16, const float* x_f32, long ldx, unsigned char* y, long ldy, float* scale,
                                    long rows, int d, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if ((x_bf16 == nullptr) == (x_f32 == nullptr) || d <= 0 || (d & 7) || d > 4096 || (ldx & 7) || (ldy & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, row_grid(rows), dim3(256), 0, s, x_bf16, x_f32, ldx, y, ldy, scale, rows, d);
    return hipGetLastError();
}
hipError_t launch_l2norm(const float* x, long ldx, float* y, long ldy, long rows, int d, float eps, int mode, hipStream_t s,
                         const float* x2) {
    if (rows <= 0) return hipSuccess;
    if (bad_width(d) || (ldx & 3) || (ldy & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(l2norm_kernel, row_grid(rows), dim3(256), 0, s, x, x2, ldx, y, ldy, rows, d, eps, mode);
    return hipGetLastError();
}
hipError_t launch_mean_rows(const float* x, long ldx, float* y, long ldy, long n, int P, int d, long group_stride, long row_add,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (bad_width(d) || (ldx & 3) || (ldy & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mean_rows_kernel, row_grid(n * ((d + 255) / 256)), dim3(256), 0, s, x, ldx, y, ldy, n, P, d, group_stride, row_add);
    return hipGetLastError();
}
hipError_t launch_ce_diag_mean(const float* logits, long ld, int n, float scale, float* row_loss, float* out, hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ce_diag_rows_kernel, row_grid(n), dim3(256), 0, s, logits, ld, n, scale, row_loss);
    hipLaunchKernelGGL(mean_scalar_kernel, dim3(1), dim3(64), 0, s, row_loss, n, out);
    return hipGetLastError();
}
hipError_t launch_gather_rows(const float* x, long ldx, float* y, long ldy, long n, int d, int group, long group_stride, long row_add,
                              const int* idx, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (bad_width(d) || (ldx & 3) || (ldy & 3) || group < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, row_grid(n), dim3(256), 0, s, x, ldx, y, ldy, n, d, group, group_stride, row_add, idx);
    return hipGetLastError();
}
hipError_t launch_gather_rows_bf16(const unsigned short* x, long ldx, float* y, long ldy, long n, int d, long row_stride, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if ((d & 3) || (ldx & 3) || (ldy & 3) || row_stride < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3((unsigned)((n * (d >> 2) + 255) / 256)), dim3(256), 0, s, x, ldx, y, ldy, n, d, row_stride);
    return hipGetLastError();
}
hipError_t launch_bert_embed(const float* cls, const float* local, const float* seq, const float* type_emb, const float* pos_emb,
                             const float* gamma, const float* beta, float* X, int B, int P, int T, int d, float eps, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (bad_width(d)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bert_embed_kernel, row_grid((long)B * (1 + P + T)), dim3(256), 0, s, cls, local, seq, type_emb, pos_emb, gamma,
                       beta, X, B, P, T, d, eps);
    return hipGetLastError();
}
// Split-K tail of the CombinerSimple hidden layer (kernels.h): one thread per (row, column); a 32-lane half wave owns one
// 32-column group and reduces it with the same xor tree as the GEMM's reduce epilogue.
__global__ __launch_bounds__(256) void splitk_relu_dot_kernel(const float* kpart, int S, long M, int N, const float* bias, const float* w2,
                                                              float* partial) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const long row = blockIdx.y;
    float v = 0.0f;
    if (col < N) {
        const float* src = kpart + row * N + col;
        for (int sl = 0; sl < S; ++sl) v += src[(long)sl * M * N];      // slices in ascending order, always
        v = fmaxf(v + bias[col], 0.0f) * w2[col];
    }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    if ((threadIdx.x & 31) == 0 && col < N) partial[row * (N / 32) + col / 32] = v;
}
// Split-K tail of a bias + residual GEMM (kernels.h): C[row][col] = (sum of the S slices, ascending) + bias[col] + R[row][col]
__global__ __launch_bounds__(256) void splitk_bias_residual_kernel(const float* kpart, int S, long M, int N, const float* bias, const float* R,
                                                                   long ldr, float* C, long ldc) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const long row = blockIdx.y;
    if (col >= N) return;
    const float* src = kpart + row * N + col;
    float v = 0.0f;
    for (int sl = 0; sl < S; ++sl) v += src[(long)sl * M * N];          // slices in ascending order, always
    v += bias ? bias[col] : 0.0f;
    if (R) v += R[row * ldr + col];
    C[row * ldc + col] = v;
}
hipError_t launch_splitk_bias_residual(const float* kpart, int S, long M, int N, const float* bias, const float* R, long ldr, float* C, long ldc,
                                       hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (S < 1 || N < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(splitk_bias_residual_kernel, dim3((N + 255) / 256, (unsigned)M), dim3(256), 0, s, kpart, S, M, N, bias, R, ldr, C, ldc);
    return hipGetLastError();
}
hipError_t launch_splitk_relu_dot(const float* kpart, int S, long M, int N, const float* bias, const float* w2, float* partial, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (S < 1 || (N & 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(splitk_relu_dot_kernel, dim3((N + 255) / 256, (unsigned)M), dim3(256), 0, s, kpart, S, M, N, bias, w2, partial);
    return hipGetLastError();
}

hipError_t launch_text_embed(const int64_t* text, const float* tok_emb, const float* pos_emb, float* X, int* eot, int B, int T, int d,
                             int vocab, int* bad_flag, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (bad_width(d)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(text_embed_kernel, row_grid((long)B * T), dim3(256), 0, s, text, tok_emb, pos_emb, X, B, T, d, vocab, bad_flag);
    hipLaunchKernelGGL(text_eot_kernel, dim3((B + 3) / 4), dim3(256), 0, s, text, eot, B, T);
    return hipGetLastError();
}
hipError_t launch_vit_cls(const float* cls, const float* pos, float* X, int B, int tokens, int d, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (bad_width(d)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vit_cls_kernel, row_grid(B), dim3(256), 0, s, cls, pos, X, B, tokens, d);
    return hipGetLastError();
}
hipError_t launch_combiner_finalize(const float* partial, int nb, const float* b2, const float* image, const float* text, float* out,
                                    long n, int d, hipStream_t s, const float* extra) {
    if (n <= 0) return hipSuccess;
    if (bad_width(d)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(combiner_finalize_kernel, row_grid(n), dim3(256), 0, s, partial, nb, b2, image, text, extra, out, n, d);
    return hipGetLastError();
}
hipError_t launch_sr_finalize(const float* partial, int nb, const float* bc, const float* local, float* out, long n, int d, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (bad_width(d)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sr_finalize_kernel, row_grid(n), dim3(256), 0, s, partial, nb, bc, local, out, n, d);
    return hipGetLastError();
}
hipError_t launch_stem_conv(const float* img, const float* w, const float* bias, float* out, int B, int S, int cout_pad, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if ((S & 1) || (cout_pad & 3)) return hipErrorInvalidValue;
    if (cout_pad == 48) {      // RN50x4's stem width (40 channels, padded to the next conv's k granularity): all channels of a pixel in one thread
        const long npix = (long)B * (S / 2) * (S / 2);
        hipLaunchKernelGGL(stem_conv_px_kernel<12>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, img, w, bias, out, B, S);
        return hipGetLastError();
    }
    const long total = (long)B * (S / 2) * (S / 2) * (cout_pad / 4);
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, w, bias, out, B, S, cout_pad);
    return hipGetLastError();
}
hipError_t launch_avgpool_nhwc(const float* x, float* y, int B, int H, int W, int C, int k, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if ((C & 3) || k < 1 || H % k || W % k) return hipErrorInvalidValue;
    const long total = (long)B * (H / k) * (W / k) * (C / 4);
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, B, H, W, C, k);
    return hipGetLastError();
}
hipError_t launch_attnpool_tokens(const float* x, float* mean, const float* pos, float* T, float* T0, int B, int HW, int C, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (C & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mean_tokens_kernel, dim3((unsigned)(((long)B * (C / 4) + 255) / 256)), dim3(256), 0, s, x, mean, B, HW, C);
    const long total = (long)B * (HW + 1) * (C / 4);
    hipLaunchKernelGGL(attnpool_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, mean, pos, T, T0, B, HW, C);
    return hipGetLastError();
}
hipError_t launch_gather_scores(const float* q, const float* gallery, const int* idx, float* out, int B, int m, int d, hipStream_t s) {
    if ((long)B * m <= 0) return hipSuccess;
    if (bad_width(d)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_scores_kernel, row_grid((long)B * m), dim3(256), 0, s, q, gallery, idx, out, B, m, d);
    return hipGetLastError();
}

}  // namespace fern
