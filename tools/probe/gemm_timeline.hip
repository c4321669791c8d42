// Per-wave timeline of the fp32 LDS-DMA GEMM kernel ("a PMC-backed stall attribution (per-wave timeline)").
// Compiles the library's own gemm.hip with FERN_GEMM_TRACE (cycle stamps after the prologue, after every k tile's barrier and
// after the epilogue; the library itself is never built with the macro) and dumps one CSV row per wave:
//     wg,wave,xcc,se,cu,simd,rt_in,rt_out,c0,c1,...          (rt_* = 100 MHz realtime, c* = shader cycle counter)
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/gemm_timeline.hip -o tools/probe/gemm_timeline
//               tools/probe/gemm_timeline M N K cfg epi out.csv        (cfg 8..11 = the LDS-DMA tile family; epi as GemmEpi)
// tools/gemm_timeline.py turns the CSV into the attribution table.
#define FERN_GEMM_TRACE 1
#include "../../fashionern_aaai2024_amd/csrc/gemm.hip"
namespace fern {      // the library defines these in api.hip; the probe never arms the launch timer
thread_local LaunchTimer* g_launch_timer = nullptr;
hipEvent_t launch_timer_event() { return nullptr; }
}  // namespace fern

#include <cmath>
#include <vector>

static hipStream_t s_;
int main(int argc, char** argv) {
    using namespace fern;
    const int M = argc > 1 ? atoi(argv[1]) : 12608, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    const int cfg = argc > 4 ? atoi(argv[4]) : 8, epi = argc > 5 ? atoi(argv[5]) : EPI_BIAS;
    const char* out = argc > 6 ? argv[6] : "gemm_timeline.csv";
    const int packed = argc > 7 ? atoi(argv[7]) : 0;      // bit 0: operands k-slab-major (timing experiment; results are then garbage); bits 8+: tile order in column stripes of that many tiles
    const int ksplit = argc > 8 ? atoi(argv[8]) : 1;      // > 1: split-K slices (raw partial sums to a scratch buffer; the reduce pass is not timed here)
    float *A, *W, *C, *bias;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&bias, (size_t)N * 4);
    std::vector<float> h((size_t)std::max(M, N) * K);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, (size_t)N * 4);
    hipMemset(C, 0, (size_t)M * N * 4);
    GemmParams p{};
    p.A = A; p.W = W; p.C = C; p.bias = bias; p.R = C; p.lda = K; p.ldw = K; p.ldc = N; p.M = M; p.N = N; p.K = K; p.epi = epi; p.aload = ALOAD_PLAIN; p.packed = packed;
    if (ksplit > 1) { p.ksplit = ksplit; hipMalloc(&p.kpart, (size_t)ksplit * M * N * 4); }
    // cfg >= 100: configuration cfg - 100 of the f32x3 family (three bf16 planes per operand); untraced timing + accuracy only
    const bool x3 = cfg >= 100;
    auto launch_any = [&](const GemmParams& q) { return x3 ? launch_cfg_split(cfg - 100, q, s_) : launch_cfg(cfg, q, s_); };
    if (x3) p.split = 3;
    const int bmv = x3 ? kCfgsS[cfg - 100].bm : kCfgs[cfg].bm, bnv = x3 ? kCfgsS[cfg - 100].bn : kCfgs[cfg].bn;
    const long nwg = (long)((M + bmv - 1) / bmv) * ((N + bnv - 1) / bnv), waves = nwg * ((cfg == 12 || cfg == 13) ? 8 : 4);
    long long* trace;
    hipMalloc(&trace, waves * FERN_GEMM_TRACE_SLOTS * 8);
    hipMemset(trace, 0, waves * FERN_GEMM_TRACE_SLOTS * 8);
    hipStream_t s;
    hipStreamCreate(&s);
    s_ = s;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    p.trace = nullptr;
    // long warm-up: the first milliseconds after an idle spell run on ramping clocks (a 5-launch warm-up measured 1.9 GHz)
    const int warm = (int)(3e11 / (2.0 * M * N * K)) + 20;
    for (int i = 0; i < warm; ++i) launch_any(p);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) launch_any(p);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("untraced: %.2f us per launch = %.1f TFLOP/s (%d x %d x %d, cfg %d, epi %d, %ld workgroups)\n", ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12,
           M, N, K, cfg, epi, nwg);
    {   // accuracy of whatever arithmetic this build runs: 512 sampled outputs against double precision (epilogue 0 only)
        std::vector<float> hc((size_t)M * N);
        hipMemset(C, 0, (size_t)M * N * 4);
        launch_any(p);
        hipStreamSynchronize(s);
        hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
        double max_err = 0, sum2 = 0;
        if (epi == EPI_BIAS && !packed) {
            for (int t = 0; t < 512; ++t) {
                const long r = (long)(rand() % M), c = (long)(rand() % N);
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)h[(size_t)r * K + k] * (double)h[(size_t)c * K + k];
                max_err = std::max(max_err, std::fabs(ref - (double)hc[(size_t)r * N + c]));
                sum2 += ref * ref;
            }
            printf("accuracy: max |error| %.3e over 512 sampled outputs, output rms %.3f -> %.2e relative\n", max_err, std::sqrt(sum2 / 512), max_err / std::sqrt(sum2 / 512));
        }
    }
    if (x3) return 0;
    p.trace = trace;
    hipEventRecord(e0, s);
    launch_cfg(cfg, p, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("traced:   %.2f us\n", ms * 1000);
    std::vector<long long> t(waves * FERN_GEMM_TRACE_SLOTS);
    hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost);
    FILE* f = fopen(out, "w");
#ifdef FERN_GEMM_TRACE_PHASES
    const int nk = K / kCfgs[cfg].bk, marks = 3 * nk + 3, phases = 3;
#else
    const int nk = K / kCfgs[cfg].bk, marks = nk + 3, phases = 1;
#endif
    fprintf(f, "# M=%d N=%d K=%d cfg=%d epi=%d nk=%d phases=%d\n", M, N, K, cfg, epi, nk, phases);
    for (long w = 0; w < waves; ++w) {
        const long long* r = &t[w * FERN_GEMM_TRACE_SLOTS];
        const unsigned hw = (unsigned)r[0];
        // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
        const int wpw = (cfg == 12 || cfg == 13) ? 8 : 4;
        fprintf(f, "%ld,%ld,%u,%u,%u,%u,%lld,%lld", w / wpw, w % wpw, (unsigned)r[1] & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, r[2], r[3]);
        for (int i = 0; i < marks && 4 + i < FERN_GEMM_TRACE_SLOTS; ++i) fprintf(f, ",%lld", r[4 + i]);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}
