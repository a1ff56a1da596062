// Probe (run on the GPU box): operand / result register layout of v_mfma_f32_16x16x32_bf16 on gfx950 and the
// accuracy of the exact three-way bf16 split (6 products) against an fp64 reference.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bf16_layout.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ unsigned pk(float a, float b) { bf16x2 v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk(a0, a1);
    float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    m = pk(r0, r1);
    float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = pk(s0, s1);
}
// A is 16 x 32 row-major, B is 32 x 16 row-major, D is 16 x 16 row-major.  Assumed layout:
//   lane (x = lane & 15, g = lane >> 4): A regs = A[x][8g + e], B regs = B[8g + e][x], D regs r = D[4g + r][x]
__global__ void probe(const float* A, const float* B, float* D, int terms) {
    const int lane = threadIdx.x, x = lane & 15, g = lane >> 4;
    u32x4 ah, am, al, bh, bm, bl;
    for (int e = 0; e < 4; ++e) {
        unsigned h, m, l;
        split_pair(A[x * 32 + 8 * g + 2 * e], A[x * 32 + 8 * g + 2 * e + 1], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
        split_pair(B[(8 * g + 2 * e) * 16 + x], B[(8 * g + 2 * e + 1) * 16 + x], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
    }
    f32x4 c = {0, 0, 0, 0};
#define MM(p, q) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p), __builtin_bit_cast(bf16x8, q), c, 0, 0, 0)
    if (terms >= 6) { MM(al, bh); MM(ah, bl); MM(am, bm); }
    if (terms >= 3) { MM(am, bh); MM(ah, bm); }
    MM(ah, bh);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + x] = c[r];
}
int main() {
    std::vector<float> A(512), B(512), D(256);
    srand(3);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 1e-3f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    for (int terms : {1, 3, 6}) {
        probe<<<1, 64>>>(dA, dB, dD, terms);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0, worst32 = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0; float f = 0.f;
                for (int k = 0; k < 32; ++k) { ref += (double)A[i * 32 + k] * B[k * 16 + j]; f = fmaf(A[i * 32 + k], B[k * 16 + j], f); }
                worst = fmax(worst, fabs(D[i * 16 + j] - ref)); scale = fmax(scale, fabs(ref)); worst32 = fmax(worst32, fabs(f - ref));
            }
        printf("terms=%d  max|err|/max|ref| = %.3e   (fp32 fmaf chain: %.3e)\n", terms, worst / scale, worst32 / scale);
    }
    return 0;
}
