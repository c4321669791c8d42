// Lab of the ping-pong 256 x 256 GEMM tile (csrc/gemm_pp.h): timing experiments by elimination and per-wave cycle stamps.
// Compiles the library's own kernel with FERN_GEMM_TRACE (the library is never built with the macro).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/pp_lab.hip -o tools/probe/pp_lab
//   tools/probe/pp_lab M N K mx dbg var [trace.csv]      (var: kernel variant; trace.csv needs a -DPP_STAMPS build)
//     mx 0: bf16, 1: block-scaled fp8.  dbg bits: 1 no staging after the prologue, 2 no LDS reads, 4 no MFMAs, 8 no vmcnt waits
//     (any bit: results are garbage, timing only).  trace.csv: cycle stamps of workgroup 0's waves (3 per phase from k tile 0:
//     after the reads retired | after the MFMAs were issued | after the phase's closing barrier).
#define FERN_GEMM_TRACE 1
#include "../../fashionern_aaai2024_amd/csrc/gemm_pp.h"
namespace fern {
thread_local LaunchTimer* g_launch_timer = nullptr;
hipEvent_t launch_timer_event() { return nullptr; }
}  // namespace fern

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    using namespace fern;
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int mx = argc > 4 ? atoi(argv[4]) : 0, dbg = argc > 5 ? atoi(argv[5]) : 0;
    const int var = argc > 6 ? atoi(argv[6]) : 0;
    const char* out = argc > 7 ? argv[7] : nullptr;
    const int es = mx ? 1 : 2;
    unsigned char *A, *W, *sa, *sw;
    float *C, *bias;
    const long sab = (long)(K / 128 + 1) * M * 4, swb = (long)(K / 128 + 1) * N * 4;
    hipMalloc(&A, (size_t)M * K * es); hipMalloc(&W, (size_t)N * K * es); hipMalloc(&sa, sab); hipMalloc(&sw, swb);
    hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&bias, (size_t)N * 4);
    srand(3);
    {
        std::vector<unsigned char> h((size_t)std::max(M, N) * K * es), hs(std::max(sab, swb));
        if (mx) { for (auto& b : h) { b = rand() & 0xFF; if ((b & 0x7F) == 0x7F) b ^= 1; if ((b & 0x78) == 0x78) b &= ~0x40; } }
        else { unsigned short* q = reinterpret_cast<unsigned short*>(h.data()); for (size_t i = 0; i < h.size() / 2; ++i) { float f = (float)rand() / RAND_MAX - 0.5f; unsigned u; memcpy(&u, &f, 4); q[i] = (unsigned short)(u >> 16); } }
        for (auto& b : hs) b = 118 + rand() % 6;
        hipMemcpy(A, h.data(), (size_t)M * K * es, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data() + 2, (size_t)N * K * es - 2, hipMemcpyHostToDevice);
        hipMemcpy(sa, hs.data(), sab, hipMemcpyHostToDevice);
        hipMemcpy(sw, hs.data(), swb, hipMemcpyHostToDevice);
    }
    hipMemset(bias, 0, (size_t)N * 4);
    GemmParams p{};
    p.Ab = reinterpret_cast<const unsigned short*>(A); p.Wb = reinterpret_cast<const unsigned short*>(W);
    p.C = C; p.bias = bias; p.lda = K; p.ldw = K; p.ldc = N; p.M = M; p.N = N; p.K = K; p.epi = EPI_BIAS; p.out_bf16 = 1;
    p.fp8 = mx ? 2 : 0; p.mxa = sa; p.mxw = sw; p.mxa_rows = M; p.mxw_rows = N; p.packed = dbg;
    const int nb = ((M + 255) / 256) * ((N + 255) / 256);
    long long* trace;
    hipMalloc(&trace, (size_t)nb * 8 * FERN_GEMM_TRACE_SLOTS * 8);
    hipMemset(trace, 0, (size_t)nb * 8 * FERN_GEMM_TRACE_SLOTS * 8);
    hipStream_t s;
    hipStreamCreate(&s);
    auto launch = [&]() {
        if (mx && var == 1) hipLaunchKernelGGL((gemm_pp_kernel<true, 4, 1>), dim3(nb), dim3(512), 0, s, p);
        else if (mx) hipLaunchKernelGGL((gemm_pp_kernel<true, 4, 0>), dim3(nb), dim3(512), 0, s, p);
        else if (var == 1) hipLaunchKernelGGL((gemm_pp_kernel<false, 4, 1>), dim3(nb), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_pp_kernel<false, 4, 0>), dim3(nb), dim3(512), 0, s, p);
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = (int)(3e11 / (2.0 * M * N * K)) + 20;
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("M %d N %d K %d %s var %d dbg %d: %.1f us  %.1f TFLOP/s  (%d workgroups)\n", M, N, K, mx ? "mx8" : "bf16", var, dbg, ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12, nb);
    if (out) {
        p.trace = trace;
        launch();
        hipStreamSynchronize(s);
        std::vector<long long> h((size_t)8 * FERN_GEMM_TRACE_SLOTS);
        hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(out, "w");
        for (int w = 0; w < 8; ++w) {
            fprintf(f, "%d", w);
            for (int i = 0; i < FERN_GEMM_TRACE_SLOTS; ++i) fprintf(f, ",%lld", h[(size_t)w * FERN_GEMM_TRACE_SLOTS + i] - h[0]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return 0;
}
