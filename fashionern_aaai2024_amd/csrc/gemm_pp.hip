// Translation unit of the ping-pong 256 x 256 GEMM tile (gemm_pp.h): configuration 7 of the bf16 family and 11 of the block-scaled
// fp8 family of gemm_bf16.hip, which owns the tile choice; kept apart so that the kernel builds on its own.
#include "gemm_pp.h"

#include <cstdlib>

namespace fern {

// FERN_PP_VAR (lab switch, read once): kernel variant -- 0: LDS-DMA staging in the load sections, 1 (default): in the MFMA sections
// (profiles/r06_pp_lab.txt: bf16 4096^3 965 -> 1083 TFLOP/s).  Every variant returns the same bits.  (A software-pipelined form with one barrier per phase --
// next phase's fragment reads between this phase's MFMAs -- was built, bit-identical and slower: profiles/r06_pp_lab.txt)
static int pp_variant() {
    static const int v = [] { const char* e = getenv("FERN_PP_VAR"); return e ? atoi(e) : 1; }();
    return v;
}

hipError_t launch_gemm_pp(bool mx, const GemmParams& p, hipStream_t s) {
    const int nb = ((p.M + PP_BM - 1) / PP_BM) * ((p.N + PP_BN - 1) / PP_BN);
    const int var = pp_variant();
    if (mx && var == 1) FERN_LAUNCH((gemm_pp_kernel<true, 4, 1>), dim3(nb), dim3(512), 0, s, p);
    else if (mx) FERN_LAUNCH((gemm_pp_kernel<true, 4, 0>), dim3(nb), dim3(512), 0, s, p);
    else if (var == 1) FERN_LAUNCH((gemm_pp_kernel<false, 4, 1>), dim3(nb), dim3(512), 0, s, p);
    else FERN_LAUNCH((gemm_pp_kernel<false, 4, 0>), dim3(nb), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace fern
