// Probe: does v_mfma_f32_16x16x4_f32 accumulate its 4 k-slots as a sequential fp32 FMA chain (slot 0,1,2,3), like a chain of
// fmaf?  And v_mfma_f32_32x32x2_f32 its 2 slots?  Compares one MFMA against fmaf chains in every slot order on random data.
// Build+run: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma16_probe.hip -o /tmp/mfma16_probe && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float* A /*[16][4]*/, const float* B /*[4][16]*/, const float* C /*[16][16]*/, float* D) {
    const int l = threadIdx.x;
    const float a = A[(l & 15) * 4 + (l >> 4)];
    const float b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
__global__ void k32(const float* A /*[32][2]*/, const float* B /*[2][32]*/, const float* C /*[32][32]*/, float* D) {
    const int l = threadIdx.x;
    const float a = A[(l & 31) * 2 + (l >> 5)];
    const float b = B[(l >> 5) * 32 + (l & 31)];
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
    float hA[64], hB[64], hC[1024], hD[1024];
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    int perms[24][4], np = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) for (int d = 0; d < 4; ++d)
        if (a != b && a != c && a != d && b != c && b != d && c != d) { perms[np][0] = a; perms[np][1] = b; perms[np][2] = c; perms[np][3] = d; ++np; }
    long match16[24] = {0}, match32[2] = {0}, fused16 = 0, total16 = 0, total32 = 0;
    srand(1);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 64; ++i) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = (float)rand() / RAND_MAX * 2 - 1; }
        for (int i = 0; i < 1024; ++i) hC[i] = ((float)rand() / RAND_MAX * 2 - 1) * (trial % 3 == 0 ? 100.f : 1.f);
        hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            ++total16;
            for (int p = 0; p < 24; ++p) {
                float acc = hC[i * 16 + j];
                for (int s = 0; s < 4; ++s) acc = fmaf(hA[i * 4 + perms[p][s]], hB[perms[p][s] * 16 + j], acc);
                if (acc == hD[i * 16 + j]) ++match16[p];
            }
            double ex = hC[i * 16 + j];
            for (int s = 0; s < 4; ++s) ex += (double)hA[i * 4 + s] * hB[s * 16 + j];
            if ((float)ex == hD[i * 16 + j]) ++fused16;
        }
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            ++total32;
            float a0 = fmaf(hA[i * 2 + 1], hB[32 + j], fmaf(hA[i * 2], hB[j], hC[i * 32 + j]));
            float a1 = fmaf(hA[i * 2], hB[j], fmaf(hA[i * 2 + 1], hB[32 + j], hC[i * 32 + j]));
            if (a0 == hD[i * 32 + j]) ++match32[0];
            if (a1 == hD[i * 32 + j]) ++match32[1];
        }
    }
    printf("32x32x2: order (0,1) matches %ld / %ld, order (1,0) %ld\n", match32[0], total32, match32[1]);
    printf("16x16x4: exactly-rounded-sum matches %ld / %ld\n", fused16, total16);
    for (int p = 0; p < 24; ++p) if (match16[p] * 100 > total16 * 90) printf("16x16x4: fmaf chain order (%d,%d,%d,%d) matches %ld / %ld\n", perms[p][0], perms[p][1], perms[p][2], perms[p][3], match16[p], total16);
    long best = 0; for (int p = 0; p < 24; ++p) if (match16[p] > best) best = match16[p];
    printf("16x16x4: best chain order matches %ld / %ld\n", best, total16);
    return 0;
}
