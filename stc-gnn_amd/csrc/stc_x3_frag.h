// Internal: split-operand bf16 MFMA helpers shared by the node kernels of stc_node_x3.hip and stc_cell_bwd_x3.hip (gfx950).
// Every fp32 operand is split EXACTLY into three bf16 pieces, a = a_h + a_m + a_l, and a product sum is accumulated in fp32 from the
// six piece products of weight >= 2^-16 on v_mfma_f32_16x16x32_bf16 (see the header of stc_node_x3.hip for the layouts).
#pragma once
#include "stc_node_frag.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct X3 { u32x4 h, m, l; };       // 8 fp32 values as three bf16x8 pieces

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {     // v_cvt_pk_bf16_f32: a in the low half
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// a - (low half of pk as a float) and a - (high half): ONE v_dot2c_f32_bf16 each (a + pk.lo * -1 + pk.hi * 0) instead of shift / mask +
// subtract.  The products and the sum are exact in fp32 (the remainder of a bf16 rounding is representable), so the pieces are the same
// bits as before; 13 -> 7 vector instructions per pair of values in kernels that are bound by vector issue.
// The constant pair must NOT reach the instruction as an inline constant: the compiler encodes {-1, 0} as the inline operand -1.0, which
// this instruction does not read as a bf16 pair on MI355X (tools/probes/dot2c_bf16.hip: r0 = 1.2422 instead of 1.9276e-4; the 32-bit
// literal 0xBF800000 and register operands are right).  Hence the constants come out of an s_mov the optimiser cannot see through.
__device__ __forceinline__ unsigned x3_pair_const(unsigned bits) {
    unsigned v;
    asm("s_mov_b32 %0, %1" : "=s"(v) : "i"(bits));
    return v;
}
__device__ __forceinline__ float minus_lo(float a, unsigned pk) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, x3_pair_const(0x0000BF80u)), a, false);
}
__device__ __forceinline__ float minus_hi(float a, unsigned pk) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, x3_pair_const(0xBF800000u)), a, false);
}

__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a0, a1);
    const float r0 = minus_lo(a0, h), r1 = minus_hi(a1, h);      // exact
    m = pk_bf16(r0, r1);
    const float s0 = minus_lo(r0, m), s1 = minus_hi(r1, m);      // exact
    l = pk_bf16(s0, s1);
}

// Eight values at once, LEVEL BY LEVEL over the four pairs: on MI355X a vector instruction that depends on the previous one issues 8 cycles
// after it, not 4, and v_cvt_pk_bf16_f32 is a half-rate instruction (tools/probes/valu_rates.hip: 8.1 cycles independent, 8.4 dependent;
// v_dot2c / v_sub / v_perm 4.8-5.3 independent, 8.4 dependent) -- one pair's chain cvt -> dot2c -> cvt -> dot2c -> cvt alone runs at the
// dependent rate, four pairs interleaved at the independent one.  The last piece needs no rounding: the second remainder has at most 8
// significant bits, so its upper 16 bits ARE its bf16 value -- one full-rate v_perm_b32 per pair instead of a conversion.
__device__ __forceinline__ X3 split8(const f32x4 a, const f32x4 b) {     // slots 0..3 from a, 4..7 from b
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    unsigned h[4], m[4], l[4];
    float r[8], s[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[2 * i] = minus_lo(v[2 * i], h[i]); r[2 * i + 1] = minus_hi(v[2 * i + 1], h[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = pk_bf16(r[2 * i], r[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[2 * i] = minus_lo(r[2 * i], m[i]); s[2 * i + 1] = minus_hi(r[2 * i + 1], m[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] = __builtin_amdgcn_perm(__float_as_uint(s[2 * i + 1]), __float_as_uint(s[2 * i]), 0x07060302u);
    X3 o;
    o.h = u32x4{h[0], h[1], h[2], h[3]};
    o.m = u32x4{m[0], m[1], m[2], m[3]};
    o.l = u32x4{l[0], l[1], l[2], l[3]};
    return o;
}

__device__ __forceinline__ f32x4 mma(const u32x4 a, const u32x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mma6(const X3& A, const X3& B, f32x4 c) {      // smallest terms first
    c = mma(A.l, B.h, c);
    c = mma(A.h, B.l, c);
    c = mma(A.m, B.m, c);
    c = mma(A.m, B.h, c);
    c = mma(A.h, B.m, c);
    c = mma(A.h, B.h, c);
    return c;
}

// LDS fragment tables: entry (frag, piece, lane) is one 16-byte vector
__device__ __forceinline__ void put_frag(u32x4* tab, int frag, int lane, const float (&v)[8]) {
    const X3 s = split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
    tab[(frag * 3 + 0) * 64 + lane] = s.h;
    tab[(frag * 3 + 1) * 64 + lane] = s.m;
    tab[(frag * 3 + 2) * 64 + lane] = s.l;
}

__device__ __forceinline__ X3 get_frag(const u32x4* tab, int frag, int lo) {
    X3 r;
    r.h = tab[(frag * 3 + 0) * 64 + lo];
    r.m = tab[(frag * 3 + 1) * 64 + lo];
    r.l = tab[(frag * 3 + 2) * 64 + lo];
    return r;
}

// row of a pair of 16-row tiles that slot (g, e) of an accumulator-fed operand stands for
__host__ __device__ constexpr int pair_row(int g, int e) { return 16 * (e >> 2) + 4 * g + (e & 3); }

#define kZero4 (f32x4{0.f, 0.f, 0.f, 0.f})

// Gate nonlinearities of the fused epilogues on the hardware exp2 / rcp (1 ulp each): absolute error < 2e-7, against the
// ~30 VALU instructions each of the IEEE division and libm expf / tanhf -- these kernels are VALU-issue bound.
__device__ __forceinline__ float fast_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float fast_tanh(float v) {       // 1 - 2 / (e^{2v} + 1); saturates cleanly at +-1
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * v));
}

}  // namespace
