// Probe (MI355X): issue cost of the vector instructions the fp32 -> 3 x bf16 operand split can be built from, one wave per SIMD,
// independent streams (8 accumulators) and one dependent chain.  s_memtime ticks = shader cycles (MI355X_MICROARCH.md).
// hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY_IND(OP) _Pragma("unroll") for (int u = 0; u < 8; ++u) { REP8(OP) }
template <int WHICH, int DEP>
__global__ void k(unsigned long long* out, float seed) {
    float a[8]; unsigned b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = __float_as_uint(a[i]) ^ 0x1234567u; }
    unsigned kc; asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(kc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 2000; ++it) {
#define I(n) (DEP ? 0 : n)
#define DOT(n) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[I(n)]) : "s"(kc), "v"(b[n]));
#define CVT(n) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b[I(n)]) : "v"(a[I(n)]), "v"(a[(n + 1) & 7]));
#define SUB(n) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[I(n)]) : "v"(a[(n + 1) & 7]));
#define AND_(n) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(b[I(n)]));
#define PERM(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(b[I(n)]) : "v"(b[(n + 1) & 7]), "s"(kc));
#define FMA(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[I(n)]) : "v"(a[(n + 1) & 7]));
#define LSHL(n) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(b[I(n)]));
        if (WHICH == 0) { BODY_IND(DOT) } else if (WHICH == 1) { BODY_IND(CVT) } else if (WHICH == 2) { BODY_IND(SUB) }
        else if (WHICH == 3) { BODY_IND(AND_) } else if (WHICH == 4) { BODY_IND(PERM) } else if (WHICH == 5) { BODY_IND(FMA) } else { BODY_IND(LSHL) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + __uint_as_float(b[i]);
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}
template <int W, int D> void run(const char* name, unsigned long long* d) {
    hipLaunchKernelGGL((k<W, D>), dim3(1024), dim3(64), 0, 0, d, 1.5f); hipDeviceSynchronize();
    hipLaunchKernelGGL((k<W, D>), dim3(1024), dim3(64), 0, 0, d, 1.5f); hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-22s %s: %6.2f cycles per instruction\n", name, D ? "dependent chain " : "8 indep. streams", (double)h[0] / (2000.0 * 64));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    run<0, 0>("v_dot2c_f32_bf16", d); run<0, 1>("v_dot2c_f32_bf16", d);
    run<1, 0>("v_cvt_pk_bf16_f32", d); run<1, 1>("v_cvt_pk_bf16_f32", d);
    run<2, 0>("v_sub_f32", d); run<2, 1>("v_sub_f32", d);
    run<3, 0>("v_and_b32", d); run<4, 0>("v_perm_b32", d); run<4, 1>("v_perm_b32", d);
    run<5, 0>("v_fma_f32", d); run<5, 1>("v_fma_f32", d); run<6, 0>("v_lshlrev_b32", d);
    return 0;
}
