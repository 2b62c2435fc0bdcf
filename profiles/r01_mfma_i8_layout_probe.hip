#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// probe: every lane sets exactly one byte of A (or B) to a code, rest zero; find the layout from the result
__global__ void probe(const int8_t* A, const int8_t* B, int* D) {
    // A: [lane][16] bytes, B: [lane][16] bytes as given by the host (lane-private operand registers)
    const int l = threadIdx.x;
    v4i a = *(const v4i*)(A + l * 16);
    v4i b = *(const v4i*)(B + l * 16);
    v16i c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}
int main() {
    // logical matrices: Am[m][k] (32x32), Bm[k][n] (32x32): asymmetric small values
    int8_t Am[32][32], Bm[32][32];
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 32; ++k) { Am[m][k] = (int8_t)((m * 3 + k * 5) % 7 - 3); Bm[m][k] = (int8_t)((m * 2 + k * 7) % 5 - 2); }
    int ref[32][32];
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int k = 0; k < 32; ++k) s += Am[m][k] * Bm[k][n]; ref[m][n] = s; }
    // hypothesis: lane l holds A[m = l&31][k = 16*(l>>5) + j], B[k = 16*(l>>5) + j][n = l&31]; D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
    std::vector<int8_t> A(64 * 16), B(64 * 16);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) { A[l * 16 + j] = Am[l & 31][16 * (l >> 5) + j]; B[l * 16 + j] = Bm[16 * (l >> 5) + j][l & 31]; }
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    std::vector<int> D(64 * 16);
    hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (D[l * 16 + r] != ref[row][col]) ++bad;
    }
    printf("layout hypothesis mismatches: %d of 1024\n", bad);
    if (bad) {   // try alternative k mapping: k = 8*(l>>5) + (j&7) + 16*(j>>3)
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) { int k = 8 * (l >> 5) + (j & 7) + 16 * (j >> 3); A[l * 16 + j] = Am[l & 31][k]; B[l * 16 + j] = Bm[k][l & 31]; }
        hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
        bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) { int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); if (D[l * 16 + r] != ref[row][col]) ++bad; }
        printf("alternative k mapping mismatches: %d of 1024\n", bad);
    }
    return 0;
}
