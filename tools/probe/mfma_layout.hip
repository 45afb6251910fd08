// Development probe (GPU box): the lane -> element maps of v_mfma_f32_32x32x16_bf16 that magent_amd/csrc/policy.hip assumes.
//   A: lane l holds A[row = l & 31][k = 8 * (l >> 5) + 0..7]      B: lane l holds B[k = 8 * (l >> 5) + 0..7][col = l & 31]
//   C: lane l, reg r holds C[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ __bf16 to_bf16(float f) { return (__bf16)f; }
__global__ void probe(const float *A, const float *B, float *C) {   // A[32][16], B[16][32], C[32][32]
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = to_bf16(A[(l & 31) * 16 + 8 * (l >> 5) + e]); b[e] = to_bf16(B[(8 * (l >> 5) + e) * 32 + (l & 31)]); }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hC[32 * 32], ref[32 * 32];
    for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) hA[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);          // small ints: exact in bf16
    for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) hB[k * 32 + j] = (float)((k * 2 + j * 7 + (k * j) % 3) % 9 - 4);   // asymmetric
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float s = 0; for (int k = 0; k < 16; k++) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32 * 32; i++) if (hC[i] != ref[i]) bad++;
    printf("mfma_f32_32x32x16_bf16 layout: %d mismatches of 1024\n", bad);
    if (bad) { for (int i = 0; i < 4; i++) { for (int j = 0; j < 8; j++) printf("%6.0f/%-6.0f", hC[i * 32 + j], ref[i * 32 + j]); printf("\n"); } }
    return bad != 0;
}
