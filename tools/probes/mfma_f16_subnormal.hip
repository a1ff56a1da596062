// Probe (MI355X): does v_mfma_f32_16x16x32_f16 keep subnormal fp16 inputs, and does the fp16 x 2 / three-product scheme reach fp32 accuracy?
// One wave: D = A . B with A[i][k], B[k][j] random in (-0.05, 0.05) (their second pieces are fp16 subnormals), compared with float64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h2 = __attribute__((ext_vector_type(2))) _Float16;
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ void split2(float a0, float a1, unsigned& h, unsigned& l) {
    unsigned k10, k01;
    asm("s_mov_b32 %0, 0x0000bc00" : "=s"(k10)); asm("s_mov_b32 %0, 0xbc000000" : "=s"(k01));
    const h2 hh = {(_Float16)a0, (_Float16)a1};
    const float r0 = __builtin_amdgcn_fdot2(hh, __builtin_bit_cast(h2, k10), a0, false), r1 = __builtin_amdgcn_fdot2(hh, __builtin_bit_cast(h2, k01), a1, false);
    const h2 ll = {(_Float16)r0, (_Float16)r1};
    h = __builtin_bit_cast(unsigned, hh); l = __builtin_bit_cast(unsigned, ll);
}
__global__ void k(const float* A, const float* B, float* D, float* D1) {     // A (16, 32) row-major, B (32, 16) row-major
    const int lane = threadIdx.x, x = lane & 15, g = lane >> 4;
    unsigned ah[4], al[4], bh[4], bl[4];
    for (int i = 0; i < 4; ++i) {
        split2(A[x * 32 + 8 * g + 2 * i], A[x * 32 + 8 * g + 2 * i + 1], ah[i], al[i]);
        split2(B[(8 * g + 2 * i) * 16 + x], B[(8 * g + 2 * i + 1) * 16 + x], bh[i], bl[i]);
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 AH = {ah[0], ah[1], ah[2], ah[3]}, AL = {al[0], al[1], al[2], al[3]}, BH = {bh[0], bh[1], bh[2], bh[3]}, BL = {bl[0], bl[1], bl[2], bl[3]};
    f4 c = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, AL), __builtin_bit_cast(h8, BH), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, AH), __builtin_bit_cast(h8, BL), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, AH), __builtin_bit_cast(h8, BH), c, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, AH), __builtin_bit_cast(h8, BH), c1, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { D[(4 * g + r) * 16 + x] = c[r]; D1[(4 * g + r) * 16 + x] = c1[r]; }
}
int main() {
    std::vector<float> A(512), B(512); std::vector<double> ref(256, 0.0);
    unsigned s = 12345; auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) / 16777216.0f - 0.5f) * 0.1f; };
    for (auto& v : A) v = rnd(); for (auto& v : B) v = rnd();
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 32; ++kk) ref[i * 16 + j] += (double)A[i * 32 + kk] * B[kk * 16 + j];
    float *dA, *dB, *dD, *dD1; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024); hipMalloc(&dD1, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD1); hipDeviceSynchronize();
    std::vector<float> D(256), D1(256); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(D1.data(), dD1, 1024, hipMemcpyDeviceToHost);
    double e3 = 0, e1 = 0, n = 0;
    for (int i = 0; i < 256; ++i) { e3 = fmax(e3, fabs(D[i] - ref[i])); e1 = fmax(e1, fabs(D1[i] - ref[i])); n = fmax(n, fabs(ref[i])); }
    printf("operands in (-0.05, 0.05): max-norm relative error  one product (hh) %.2e   three products (lh + hl + hh) %.2e\n", e1 / n, e3 / n);
    return 0;
}
