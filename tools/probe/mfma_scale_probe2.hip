// Second probe of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3): WHICH of a lane's bytes does each lane's scale apply to, and which B byte
// does A's byte (h, j) multiply?  (mfma_scale_probe.hip used all-ones data for the scale test, which cannot tell.)
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_scale_probe2.hip -o /tmp/mx_probe2 && /tmp/mx_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* areg, const unsigned char* breg, const int* sa, const int* sb, float* D) {
    const int l = threadIdx.x;
    i32x8 a, b;
    memcpy(&a, areg + l * 32, 32);
    memcpy(&b, breg + l * 32, 32);
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
static unsigned char f8(int v) {
    static const unsigned char pos[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return pos[v];
}
int main() {
    unsigned char hA[2048], hB[2048];
    int hsa[64], hsb[64];
    float hD[1024];
    unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 4096);
    auto run = [&]() {
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    };
    // 1. A one-hot at (row 0, half h, byte j); B all ones; scale_a: lane 0 -> x2, lane 32 -> x4, others 1; which applies?
    printf("A byte (h, j) -> scale of lane half that applies (0 / 1):\n");
    for (int h = 0; h < 2; ++h) {
        printf("  h=%d: ", h);
        for (int j = 0; j < 32; ++j) {
            memset(hA, 0, 2048); memset(hB, 0x38, 2048);
            hA[(32 * h) * 32 + j] = 0x38;
            for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
            hsa[0] = 128; hsa[32] = 129;
            run();
            printf("%c", hD[0] == 2.f ? '0' : hD[0] == 4.f ? '1' : '?');
        }
        printf("\n");
    }
    printf("B byte (h, j) -> scale_b of lane half that applies:\n");
    for (int h = 0; h < 2; ++h) {
        printf("  h=%d: ", h);
        for (int j = 0; j < 32; ++j) {
            memset(hB, 0, 2048); memset(hA, 0x38, 2048);
            hB[(32 * h) * 32 + j] = 0x38;
            for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
            hsb[0] = 128; hsb[32] = 129;
            run();
            printf("%c", hD[0] == 2.f ? '0' : hD[0] == 4.f ? '1' : '?');
        }
        printf("\n");
    }
    // 2. pairing: A one-hot (row 0, h, j); B (col 0) byte (h', j') = value encoding; D[0][0] decodes the B byte it met
    printf("A byte (h, j) multiplies B byte (h', j'):\n");
    for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 32; ++j) {
            int dec[2];
            for (int pass = 0; pass < 2; ++pass) {
                memset(hA, 0, 2048);
                hA[(32 * h) * 32 + j] = 0x38;
                for (int l = 0; l < 64; ++l) { hsa[l] = hsb[l] = 127; }
                for (int l = 0; l < 64; ++l) for (int jj = 0; jj < 32; ++jj) {
                    const int id = (l >> 5) * 32 + jj;
                    hB[l * 32 + jj] = f8(pass == 0 ? (id & 7) + 1 : (id >> 3) + 1);
                }
                run();
                dec[pass] = (int)hD[0] - 1;
            }
            const int id = dec[0] + 8 * dec[1];
            printf("  (%d,%2d)->(%d,%2d)%s", h, j, id >> 5, id & 31, (j & 7) == 7 ? "\n" : "");
        }
    return 0;
}
