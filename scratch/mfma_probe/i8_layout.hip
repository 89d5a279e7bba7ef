// Probe of the operand lane map of v_mfma_i32_16x16x64_i8 with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const i32x4 *a, const i32x4 *b, i32x4 *d)
{
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}
int main()
{
    int8_t A[16][64], B[64][16];
    srand(1);
    for (int r = 0; r < 16; ++r) for (int kk = 0; kk < 64; ++kk) A[r][kk] = (rand() % 15) - 7;
    for (int kk = 0; kk < 64; ++kk) for (int c = 0; c < 16; ++c) B[kk][c] = (rand() % 15) - 7;
    int ref[16][16];
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { int s = 0; for (int kk = 0; kk < 64; ++kk) s += A[r][kk] * B[kk][c]; ref[r][c] = s; }
    for (int hyp = 0; hyp < 3; ++hyp) {
        int8_t ha[64][16], hb[64][16];
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
            int kk;
            if (hyp == 0) kk = 16 * (l >> 4) + j;                       // contiguous 16 per k-group
            else if (hyp == 1) kk = (j < 8) ? 8 * (l >> 4) + j : 32 + 8 * (l >> 4) + (j - 8); // two K=32 halves
            else kk = 4 * (l >> 4) + (j & 3) + 16 * (j >> 2);           // four K=16 quarters
            ha[l][j] = A[l & 15][kk];
            hb[l][j] = B[kk][l & 15];
        }
        i32x4 *da, *db, *dd;
        hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
        hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
        int out[64][4];
        hipMemcpy(out, dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) bad += out[l][i] != ref[4 * (l >> 4) + i][l & 15];
        printf("hypothesis %d: %d mismatches (D map col = lane & 15, row = 4 * (lane >> 4) + reg)\n", hyp, bad);
    }
    return 0;
}
