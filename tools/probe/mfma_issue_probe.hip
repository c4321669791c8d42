// How fast can v_mfma_f32_32x32x2_f32 issue?  One workgroup per CU, W waves per SIMD, each wave runs back-to-back MFMAs on NACC
// independent accumulators and no memory traffic at all: cycles per MFMA per SIMD = 64 means the pipe is saturated.
// Variants: W = 1, 2, 4 waves per SIMD; NACC = 1, 2, 4; with a v_add / ds_read sprinkled in (the GEMM loop's side instructions).
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_issue_probe.hip -o tools/probe/mfma_issue_probe && tools/probe/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int SIDE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    __shared__ float lds[4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    int side = threadIdx.x;
    lds[threadIdx.x] = a;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (SIDE == 1) side = side * 3 + u;                                  // one VALU op per NACC MFMAs
            if (SIDE == 2) a += lds[(side + u * 64) & 4095] * 1e-9f;             // one LDS read + dependent VALU per NACC MFMAs
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = (float)side;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int SIDE>
void run(int waves_per_simd) {
    float* out; long long* cyc;
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL((k<NACC, SIDE>), dim3(blocks), dim3(threads), 0, 0, out, 10, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SIDE>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    const double mfma_per_simd = (double)iters * 16 * NACC * waves_per_simd;
    printf("waves/SIMD %d  accumulators %d  side %d: %.1f cycles per MFMA per SIMD  (%.1f TFLOP/s chip)\n", waves_per_simd, NACC, SIDE, avg / mfma_per_simd,
           mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) { run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<4, 1>(w); run<4, 2>(w); }
    return 0;
}
