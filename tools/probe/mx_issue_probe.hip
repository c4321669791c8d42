// How fast does v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands) issue, and what does the block-scaled GEMM's k loop lose around it?
// One workgroup per CU, W waves per SIMD, no global traffic.  Cycles per MFMA per SIMD = 64 means the pipe is saturated (16 passes).
//   MODE 0: operands in registers, NACC independent accumulators, back to back
//   MODE 1: the k loop's shape for a 64x64 wave tile: per 64-k step 8 ds_read_b128 (A 2x2, W 2x2 fragments) + 2 scale dwords from LDS,
//           s_waitcnt lgkmcnt(0), 4 MFMAs -- no barrier
//   MODE 2: MODE 1 + a workgroup barrier per 128 k (every second step), as the library kernel has
//   MODE 3: MODE 2 with the NEXT step's fragments read while this step's MFMAs issue (double-buffered fragments)
//   DATA 0: operand bytes zero (cool pipe), 1: pseudo-random e4m3 bytes (the power the real kernel draws)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/mx_issue_probe.hip -o tools/probe/mx_issue_probe && tools/probe/mx_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ i32x8 cat(i32x4 a, i32x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

template <int MODE, int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc, int data) {
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15;
        reinterpret_cast<unsigned*>(lds)[i] = data ? (h & 0x7e7e7e7eu) : 0u;      // finite e4m3 bytes (0x7f is NaN)
    }
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    // 128-byte rows, chunk swizzle of the library kernel; a wave's A rows and W rows live in its own 16 KiB quarter (4 waves share one)
    const char* base = lds + (wave & 3) * 16384;
    long long t0, t1;
    if (MODE == 0) {
        i32x8 a = cat(*reinterpret_cast<const i32x4*>(base + lane * 16), *reinterpret_cast<const i32x4*>(base + 1024 + lane * 16));
        i32x8 b = cat(*reinterpret_cast<const i32x4*>(base + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(base + 3072 + lane * 16));
        const int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb);
        }
        t1 = __builtin_readcyclecounter();
    } else {
        i32x8 af[2][2], bf[2][2];
        int sa[2][2], sb[2][2];
        auto frags = [&](int buf, int e, int it) {
            const int pc0 = ((4 * e + lh) ^ sw) * 16, pc1 = ((4 * e + 2 + lh) ^ sw) * 16;
            const char* A = base + ((it & 1) * 8192);                     // 64 rows x 128 B
            const char* W = A + 32768 > lds + 65536 - 8192 ? A : A;        // same quarter: the probe only needs the access shape
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* r = A + (i * 32 + l31) * 128;
                af[buf][i] = cat(*reinterpret_cast<const i32x4*>(r + pc0), *reinterpret_cast<const i32x4*>(r + pc1));
                const char* q = W + ((1 - i) * 32 + l31) * 128;
                bf[buf][i] = cat(*reinterpret_cast<const i32x4*>(q + pc0), *reinterpret_cast<const i32x4*>(q + pc1));
                sa[buf][i] = 0x7f7f7f7f;
                sb[buf][i] = (int)(*reinterpret_cast<const unsigned*>(lds + 4 * (i * 32 + l31)) | 0x7f7f7f7fu) >> (8 * lh) | 0x7f;
            }
        };
        auto mfmas = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[buf][i], bf[buf][j], acc[i * 2 + j], 0, 0, 0, sa[buf][i], 0, sb[buf][j]);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i]) : : "memory");
        };
        t0 = __builtin_readcyclecounter();
        if (MODE == 3) frags(0, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (MODE >= 2 && e == 0) __builtin_amdgcn_s_barrier();
                if (MODE == 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    frags(1 - e, 1 - e, it + e);
                    mfmas(e);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    frags(0, e, it);
                    mfmas(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NACC>
void run(int waves_per_simd, int data) {
    float* out; long long* cyc;
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 1000;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, 10, cyc, data);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc, data);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    const double per_wave = MODE == 0 ? (double)iters * 8 * NACC : (double)iters * 8;
    const double mfma_per_simd = per_wave * waves_per_simd;
    printf("mode %d  waves/SIMD %d  acc %d  data %s: %6.1f cycles per MFMA per SIMD   %.2f GHz   %.0f TFLOP/s chip\n", MODE, waves_per_simd, MODE == 0 ? NACC : 4,
           data ? "random" : "zero  ", avg / mfma_per_simd, avg / (ms * 1e3) * 1e-3, mfma_per_simd * 1024 * (2.0 * 32 * 32 * 64) / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int data : {0, 1}) {
        for (int w : {1, 2, 4}) { run<0, 1>(w, data); run<0, 2>(w, data); run<0, 4>(w, data); }
        for (int w : {1, 2, 4}) { run<1, 4>(w, data); run<2, 4>(w, data); run<3, 4>(w, data); }
    }
    return 0;
}
