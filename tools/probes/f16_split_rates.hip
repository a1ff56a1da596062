// Probe (MI355X): the instructions an fp32 -> 2 x fp16 operand split can be built from -- issue cost (one wave per SIMD, 8 independent
// streams / one dependent chain, s_memtime ticks = shader cycles) and exactness of the split built from them, with and without a
// power-of-two scale folded in (v_fma_mixlo/hi_f16: scaled conversion in one instruction per value; v_fma_mix_f32: a * s - h exactly).
// hipcc --offload-arch=gfx950 -O3 tools/probes/f16_split_rates.hip -o /tmp/f16_split_rates && /tmp/f16_split_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY_IND(OP) _Pragma("unroll") for (int u = 0; u < 8; ++u) { REP8(OP) }
template <int WHICH, int DEP>
__global__ void k(unsigned long long* out, float seed) {
    float a[8]; unsigned b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = __float_as_uint(a[i]) ^ 0x1234567u; }
    float sc; asm volatile("s_mov_b32 %0, 0x3f800000" : "=s"(sc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 2000; ++it) {
#define I(n) (DEP ? 0 : n)
#define CVT(n) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(b[I(n)]) : "v"(a[I(n)]), "v"(a[(n + 1) & 7]));
#define RTZ(n) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(b[I(n)]) : "v"(a[I(n)]), "v"(a[(n + 1) & 7]));
#define MIX(n) asm volatile("v_fma_mix_f32 %0, %0, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(a[I(n)]) : "s"(sc), "v"(b[(n + 1) & 7]));
#define MIXLO(n) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(b[I(n)]) : "v"(a[I(n)]), "s"(sc));
#define MIXHI(n) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(b[I(n)]) : "v"(a[I(n)]), "s"(sc));
#define DOT(n) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[I(n)]) : "v"(b[(n + 2) & 7]), "v"(b[n]));
#define MUL(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[I(n)]) : "s"(sc));
        if (WHICH == 0) { BODY_IND(CVT) } else if (WHICH == 1) { BODY_IND(RTZ) } else if (WHICH == 2) { BODY_IND(MIX) }
        else if (WHICH == 3) { BODY_IND(MIXLO) } else if (WHICH == 4) { BODY_IND(MIXHI) } else if (WHICH == 5) { BODY_IND(DOT) } else { BODY_IND(MUL) }
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

// ---- the split itself: (h, l) packed pairs of two values, unscaled and scaled
using h2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& l) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a0), "v"(a1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(a1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
}
__device__ __forceinline__ void split2s(float a0, float a1, float s, unsigned& h, unsigned& l) {
    h = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(s));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(s), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(s), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
}
__global__ void ksplit(const float* in, unsigned* out, float s, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h, l, hs, ls;
    split2(in[2 * i], in[2 * i + 1], h, l);
    split2s(in[2 * i], in[2 * i + 1], s, hs, ls);
    out[4 * i] = h; out[4 * i + 1] = l; out[4 * i + 2] = hs; out[4 * i + 3] = ls;
}
static float h2f(unsigned short v) {      // fp16 bits -> float (subnormals included)
    const int s = v >> 15, e = (v >> 10) & 31, m = v & 1023;
    float r = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? INFINITY : ldexpf((float)(m | 1024), e - 25));
    return s ? -r : r;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    run<0, 0>("v_cvt_pk_f16_f32", d); run<0, 1>("v_cvt_pk_f16_f32", d);
    run<1, 0>("v_cvt_pkrtz_f16_f32", d); run<1, 1>("v_cvt_pkrtz_f16_f32", d);
    run<2, 0>("v_fma_mix_f32", d); run<2, 1>("v_fma_mix_f32", d);
    run<3, 0>("v_fma_mixlo_f16", d); run<3, 1>("v_fma_mixlo_f16", d);
    run<4, 0>("v_fma_mixhi_f16", d); run<4, 1>("v_fma_mixhi_f16", d);
    run<5, 0>("v_dot2c_f32_f16", d); run<5, 1>("v_dot2c_f32_f16", d);
    run<6, 0>("v_mul_f32", d); run<6, 1>("v_mul_f32", d);
    const int n = 1 << 16; std::vector<float> in(n);
    unsigned sd = 777; auto rnd = [&] { sd = sd * 1664525u + 1013904223u; return (sd >> 8) / 16777216.0f; };
    for (int i = 0; i < n; ++i) in[i] = ldexpf((rnd() - 0.5f) * 2.f, (int)(rnd() * 12) - 10);      // magnitudes 2^-10 .. 2
    const float s = 64.f;
    float* din; unsigned* dout; hipMalloc(&din, n * 4); hipMalloc(&dout, n * 8);
    hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ksplit, dim3(n / 2 / 256), dim3(256), 0, 0, din, dout, s, n); hipDeviceSynchronize();
    std::vector<unsigned> o(2 * n); hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    double worst = 0, worst_s = 0, worst_abs = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int p = 0; p < 2; ++p) {
            const float a = in[2 * i + p];
            const float h = h2f((o[4 * i] >> (16 * p)) & 0xffff), l = h2f((o[4 * i + 1] >> (16 * p)) & 0xffff);
            const float hs = h2f((o[4 * i + 2] >> (16 * p)) & 0xffff), ls = h2f((o[4 * i + 3] >> (16 * p)) & 0xffff);
            if (a != 0) { worst = fmax(worst, fabs(((double)h + l - a) / a)); worst_s = fmax(worst_s, fabs(((double)hs + ls - (double)a * s) / ((double)a * s))); }
            worst_abs = fmax(worst_abs, fabs((double)h + l - a));
        }
    printf("split (h + l - a) / a, magnitudes 2^-10..2: unscaled worst %.3e (abs %.3e)   scaled by 64 worst %.3e   (2^-22 = %.3e)\n", worst, worst_abs, worst_s, ldexp(1.0, -22));
    return 0;
}
