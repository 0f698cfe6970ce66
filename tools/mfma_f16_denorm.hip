// Does v_mfma_f32_16x16x32_f16 treat fp16 SUBNORMAL inputs exactly?  A = diag-like pattern with subnormal values, B = ones / subnormals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *a, const _Float16 *b, float *d) {
    const int lane = threadIdx.x;
    h8 av, bv;
    for (int j = 0; j < 8; j++) {
        av[j] = a[(lane & 15) * 32 + 8 * (lane >> 4) + j];   // A[row = lane & 15][k]
        bv[j] = b[(lane & 15) * 32 + 8 * (lane >> 4) + j];   // B[k][col = lane & 15] stored as [col][k]
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}
int main() {
    _Float16 ha[16 * 32], hb[16 * 32];
    double ref[16][16];
    for (int i = 0; i < 16; i++)
        for (int kk = 0; kk < 32; kk++) {
            // subnormal fp16 values: 2^-24 * m, m = 1..1023
            ha[i * 32 + kk] = (_Float16)(ldexp((double)(1 + (i * 37 + kk * 11) % 1023), -24));
            hb[i * 32 + kk] = (_Float16)(kk % 3 == 0 ? ldexp((double)(1 + (i * 13 + kk * 7) % 1023), -24) : 1.0 + 0.125 * ((i + kk) % 5));
        }
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int kk = 0; kk < 32; kk++) s += (double)ha[i * 32 + kk] * (double)hb[j * 32 + kk];
            ref[i][j] = s;
        }
    _Float16 *da, *db; float *dd;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, 256 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    float out[256];
    hipMemcpy(out, dd, sizeof(out), hipMemcpyDeviceToHost);
    double maxrel = 0; int zeros = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        const double r = ref[i][j], g = out[i * 16 + j];
        if (g == 0.0) zeros++;
        const double rel = fabs(g - r) / fabs(r);
        if (rel > maxrel) maxrel = rel;
    }
    printf("subnormal-input MFMA: max relative error vs exact %.3e, zero outputs %d of 256 (ref[0][0]=%.6e got %.6e)\n", maxrel, zeros, ref[0][0], out[0]);
    return 0;
}
