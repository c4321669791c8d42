// 8-bit RGB resampling and tensor conversion for the image side of the path:
// the reference builds its inputs with PIL -- `Image.resize((360, 360), ANTIALIAS)` + 2x2 / 3x3 crops
// (utils/extract_fashioniq_patch.py:142-149) and torchvision's Resize(BICUBIC) / CenterCrop / ToTensor / Normalize on PIL
// images (dataloader/dataset.py:73-87).  The arithmetic of PIL's resize lives in third-party Pillow
// (src/libImaging/Resample.c): a separable convolution with per-output-pixel windows whose coefficients are rounded to
// fixed point (2^-22), an int32 accumulation started at 2^21, an arithmetic shift and a clamp to 0..255, with the
// intermediate image rounded to 8 bits between the horizontal and the vertical pass.  The windows and coefficients are
// computed on the host (fashionern_aaai2024_amd/preprocess.py restates precompute_coeffs / normalize_coeffs_8bpc); these
// kernels do the integer accumulation, so the result is bit-identical to PIL.
#include "kernels.h"

namespace fern {

constexpr int PRECISION_BITS = 22;

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y, ox, c] = clip8(2^21 + sum_t src[y0 + y, x0 + bounds[ox].min + t, c] * kk[ox][t])
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* src, long src_ld /*pixels per row*/, int x0, int y0, int rows,
                                                         unsigned char* dst, int ow, const int* bounds, const int* kk, int ksize) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * ow) return;
    const int ox = (int)(t % ow), y = (int)(t / ow);
    const int xmin = bounds[2 * ox], n = bounds[2 * ox + 1];
    const unsigned char* p = src + ((long)(y0 + y) * src_ld + x0 + xmin) * 3;
    const int* k = kk + (long)ox * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        s0 += p[3 * i] * w;
        s1 += p[3 * i + 1] * w;
        s2 += p[3 * i + 2] * w;
    }
    unsigned char* o = dst + ((long)y * ow + ox) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// dst[oy, x, c] = clip8(2^21 + sum_t src[y0 + bounds[oy].min + t, x0 + x, c] * kk[oy][t])
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* src, long src_ld, int x0, int y0, int cols, unsigned char* dst,
                                                         int oh, const int* bounds, const int* kk, int ksize) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)oh * cols) return;
    const int x = (int)(t % cols), oy = (int)(t / cols);
    const int ymin = bounds[2 * oy], n = bounds[2 * oy + 1];
    const unsigned char* p = src + ((long)(y0 + ymin) * src_ld + x0 + x) * 3;
    const int* k = kk + (long)oy * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        const unsigned char* q = p + (long)i * src_ld * 3;
        s0 += q[0] * w;
        s1 += q[1] * w;
        s2 += q[2] * w;
    }
    unsigned char* o = dst + ((long)oy * cols + x) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// ToTensor + Normalize (dataloader/dataset.py:84-86): HWC u8 -> CHW f32, (v / 255 - mean[c]) / std[c] with a crop window
__global__ __launch_bounds__(256) void u8_to_chw_kernel(const unsigned char* src, long src_ld, int x0, int y0, float* dst, int n, long src_img_stride,
                                                        int oh, int ow, float m0, float m1, float m2, float d0, float d1, float d2) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n * oh * ow) return;
    const int x = (int)(t % ow), y = (int)((t / ow) % oh), i = (int)(t / ((long)ow * oh));
    const unsigned char* p = src + (long)i * src_img_stride + ((long)(y0 + y) * src_ld + x0 + x) * 3;
    float* o = dst + (long)i * 3 * oh * ow + (long)y * ow + x;
    const long plane = (long)oh * ow;
    o[0] = ((float)p[0] / 255.0f - m0) / d0;
    o[plane] = ((float)p[1] / 255.0f - m1) / d1;
    o[2 * plane] = ((float)p[2] / 255.0f - m2) / d2;
}

hipError_t launch_resample_h(const unsigned char* src, long src_ld, int x0, int y0, int rows, unsigned char* dst, int ow, const int* bounds,
                             const int* kk, int ksize, hipStream_t s) {
    if (rows <= 0 || ow <= 0) return hipSuccess;
    const long total = (long)rows * ow;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_ld, x0, y0, rows, dst, ow, bounds, kk, ksize);
    return hipGetLastError();
}
hipError_t launch_resample_v(const unsigned char* src, long src_ld, int x0, int y0, int cols, unsigned char* dst, int oh, const int* bounds,
                             const int* kk, int ksize, hipStream_t s) {
    if (cols <= 0 || oh <= 0) return hipSuccess;
    const long total = (long)oh * cols;
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_ld, x0, y0, cols, dst, oh, bounds, kk, ksize);
    return hipGetLastError();
}
hipError_t launch_u8_to_chw(const unsigned char* src, long src_ld, int x0, int y0, float* dst, int n, long src_img_stride, int oh, int ow,
                            const float* mean, const float* stdv, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long total = (long)n * oh * ow;
    hipLaunchKernelGGL(u8_to_chw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_ld, x0, y0, dst, n, src_img_stride, oh, ow,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}

}  // namespace fern
