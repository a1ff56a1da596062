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

__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a0, a1);
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);      // exact
    m = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);      // exact
    l = pk_bf16(s0, s1);
}

__device__ __forceinline__ X3 split8(const f32x4 a, const f32x4 b) {     // slots 0..3 from a, 4..7 from b
    unsigned h[4], m[4], l[4];
    split2(a[0], a[1], h[0], m[0], l[0]);
    split2(a[2], a[3], h[1], m[1], l[1]);
    split2(b[0], b[1], h[2], m[2], l[2]);
    split2(b[2], b[3], h[3], m[3], l[3]);
    X3 r;
    r.h = u32x4{h[0], h[1], h[2], h[3]};
    r.m = u32x4{m[0], m[1], m[2], m[3]};
    r.l = u32x4{l[0], l[1], l[2], l[3]};
    return r;
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
