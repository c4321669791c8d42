// v_mfma_f32_32x32x16_bf16 issue rate, bare and in the f32x3 k loop's shape (fat wave 128x64: 18 ds_read_b128 + 48 MFMAs per 16-k tile).
//   MODE 0: operands in registers, NACC accumulators
//   MODE 1: per tile 18 ds_read_b128 (4 + 2 row blocks x 3 planes), lgkmcnt(0), 48 MFMAs in the family's order, no barrier
//   MODE 2: MODE 1 + a workgroup barrier per tile
//   MODE 3: MODE 2 + 72 VALU instructions of split work per tile (perm / and / packed subtract on dummy registers) + 6 ds_write_b128
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/bf16_issue_probe.hip -o tools/probe/bf16_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8s __attribute__((ext_vector_type(8)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int data) {
    __shared__ __attribute__((aligned(1024))) char lds[98304];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 98304 / 4; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15;
        reinterpret_cast<unsigned*>(lds)[i] = data ? ((h & 0x007f007fu) | 0x3f803f80u) ^ (h & 0x80008000u) : 0u;      // bf16 values in +-[1, 2)
    }
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    if (MODE == 0) {
        const bf16x8s a = *reinterpret_cast<const bf16x8s*>(lds + lane * 16), b = *reinterpret_cast<const bf16x8s*>(lds + 4096 + lane * 16);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    } else {
        constexpr int ROWS = 512, HALF = ROWS * 16, PLANE = 2 * HALF, BUF = 3 * PLANE;
        const int wm = wave >> 2, wn = wave & 3;
        const int arow = lh * HALF + (wm * 128 + l31) * 16, wrow = lh * HALF + (256 + wn * 64 + l31) * 16;
        f32x2s x[8];
        for (int q = 0; q < 8; ++q) x[q] = f32x2s{1.0f + tid * 1e-3f + q, 2.0f - tid * 1e-3f};
        for (int it = 0; it < iters; ++it) {
            const char* P = lds + (it & 1) * BUF;
            bf16x8s ap[4][3], bp[2][3];
#pragma unroll
            for (int lv = 0; lv < 3; ++lv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ap[i][lv] = *reinterpret_cast<const bf16x8s*>(P + lv * PLANE + arow + i * 512);
#pragma unroll
                for (int j = 0; j < 2; ++j) bp[j][lv] = *reinterpret_cast<const bf16x8s*>(P + lv * PLANE + wrow + j * 512);
            }
#pragma unroll
            for (int ord = 0; ord < 3; ++ord)
#pragma unroll
                for (int la = 0; la <= ord; ++la)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][la], bp[j][ord - la], acc[i * 2 + j], 0, 0, 0);
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x2s y[4] = {x[4 * u], x[4 * u + 1], x[4 * u + 2], x[4 * u + 3]};
#pragma unroll
                    for (int lv = 0; lv < 3; ++lv) {
                        u32x4s packed;
#pragma unroll
                        for (int q2 = 0; q2 < 4; ++q2) packed[q2] = __builtin_amdgcn_perm(__float_as_uint(y[q2][1]), __float_as_uint(y[q2][0]), 0x07060302u);
                        *reinterpret_cast<u32x4s*>(lds + ((it + 1) & 1) * BUF + lv * PLANE + (tid + 512 * u) * 16) = packed;
                        if (lv < 2)
#pragma unroll
                            for (int q2 = 0; q2 < 4; ++q2) {
                                const u32x2s top = __builtin_bit_cast(u32x2s, y[q2]) & 0xFFFF0000u;
                                y[q2] = y[q2] - __builtin_bit_cast(f32x2s, top);
                            }
                    }
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) x[4 * u + q2] = y[q2] + x[4 * u + q2] * 1.0001f;
                }
            }
            if (MODE >= 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(int threads, int data) {
    float* out;
    const int blocks = 256, iters = 2000;
    (void)hipMalloc(&out, (size_t)blocks * threads * 4);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, 10, data);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters, data);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = MODE == 0 ? (double)iters * 6 * NACC : (double)iters * 48;
    const double mfma = per_wave * (threads / 64) * blocks;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("mode %d  waves/SIMD %d  acc %d  data %s: %7.0f TFLOP/s bf16 = %5.1f TFLOP/s of f32x3 work (six products)   %4.1f %% of 2.5 PF\n", MODE, threads / 256,
           MODE == 0 ? NACC : 8, data ? "random" : "zero  ", tf, tf / 6.0, tf / 2500 * 100);
    (void)hipFree(out);
}
int main() {
    for (int data : {0, 1}) {
        run<0, 4>(256, data); run<0, 8>(256, data); run<0, 4>(512, data); run<0, 8>(512, data);
        run<1, 8>(512, data); run<2, 8>(512, data); run<3, 8>(512, data);
        run<1, 8>(256, data); run<2, 8>(256, data);
    }
    return 0;
}
