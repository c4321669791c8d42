// wave_sum / wave_max on the DPP network (elem.hip) against a serial sum: hipcc --offload-arch=gfx950 -O3 tools/probe/dpp_reduce_probe.hip -o /tmp/dpp && /tmp/dpp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }   // (readlane is an int builtin)
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__global__ void k(const float* a, float* o) {
    const float v = a[threadIdx.x];
    o[threadIdx.x] = wave_sum(v);
    o[64 + threadIdx.x] = wave_sum(v * v);
}
int main() {
    float h[64], r[128], *d, *o;
    double s = 0, q = 0;
    for (int i = 0; i < 64; ++i) { h[i] = (float)(i * 7 % 13) - 6.f + 0.25f * i; s += h[i]; q += (double)h[i] * h[i]; }
    hipMalloc(&d, 256); hipMalloc(&o, 512);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    printf("sum: lane0 %g lane17 %g lane63 %g expected %g\n", r[0], r[17], r[63], s);
    printf("sumsq: lane0 %g lane40 %g expected %g\n", r[64], r[104], q);
    return 0;
}
