// Probe (MI355X): semantics of v_dot2c_f32_bf16 with an inline-constant operand vs a register operand.
// hipcc --offload-arch=gfx950 -O3 tools/probes/dot2c_bf16.hip -o /tmp/dot2c && /tmp/dot2c
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__global__ void k(const float* a, const unsigned* cpair, float* out) {
    const float a0 = a[0], a1 = a[1];
    const bf16x2 h = {(__bf16)a0, (__bf16)a1};
    const unsigned hu = __builtin_bit_cast(unsigned, h);
    out[0] = __builtin_amdgcn_fdot2_f32_bf16(h, bf16x2{(__bf16)-1.0f, (__bf16)0.0f}, a0, false);      // inline constant pair {-1, 0}
    out[1] = __builtin_amdgcn_fdot2_f32_bf16(h, bf16x2{(__bf16)0.0f, (__bf16)-1.0f}, a1, false);      // literal pair {0, -1}
    out[2] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, cpair[0]), a0, false);     // the same pairs from a register
    out[3] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, cpair[1]), a1, false);
    out[4] = a0 - __uint_as_float(hu << 16);
    out[5] = a1 - __uint_as_float(hu & 0xffff0000u);
}
int main() {
    float ha[2] = {1.2345678f, -7.654321e-3f};
    unsigned hc[2] = {0x0000BF80u, 0xBF800000u};
    float *da, *dout; unsigned* dc;
    hipMalloc(&da, 8); hipMalloc(&dc, 8); hipMalloc(&dout, 24);
    hipMemcpy(da, ha, 8, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, da, dc, dout);
    float ho[6]; hipMemcpy(ho, dout, 24, hipMemcpyDeviceToHost);
    printf("inline  : r0 = %.9g  r1 = %.9g\nregister: r0 = %.9g  r1 = %.9g\nshift/sub: r0 = %.9g  r1 = %.9g\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
    return 0;
}
