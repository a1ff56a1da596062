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

// ---------------------------------------------------------------------------------------------------------------------------------
// Second operand format: TWO fp16 pieces per fp32 value, a s = a_h + a_l (11 + 11 significant bits, s a power of two chosen per operand
// class), and THREE piece products a_l b_h + a_h b_l + a_h b_h on v_mfma_f32_16x16x32_f16 -- half the matrix instructions of the bf16 x 3
// scheme and 4-5 instead of 7 vector instructions per split pair of values.  fp16 x fp16 products are exact in the fp32 accumulator; what
// is dropped is a_l b_l and the second rounding, <= 2^-22 |a||b| each (tools/probes/operand_format_error.py: 1.7-2.9e-7 on 32-term dot
// products against float64, the fp32 fmaf chain of the reference 1.3-2.4e-7).  fp16 has 5 exponent bits: the low piece of a value below
// 2^-3 is a subnormal, i.e. the pair carries an ABSOLUTE error floor of 2^-25, and values above 65504 do not exist.  Hence the scales:
//   tables (W, T_c)     normalised per workgroup to a maximum in [1/2, 1) at table-fill time (max taken in the kernel, no host round trip);
//   gradient operands   scaled into [2^3, 2^4) from the launch's gradient maximum, which the producing kernel leaves in device memory;
//   activations         bounded by construction (gate outputs, states in (-1, 1), aggregates of those); an optional bound scales them DOWN only.
// All scales are powers of two (exact), products of scaled operands are unscaled on the fp32 result.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct X2 { u32x4 h, l; };          // 8 fp32 values as two fp16x8 pieces

// 2^k with amax 2^k in [2^(t-1), 2^t); 1 for zero / subnormal / non-finite amax; |k| <= 100
__device__ __forceinline__ float pow2_scale(float amax, int t) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 255u);          // amax in [2^(e-127), 2^(e-126))
    if (e == 0 || e == 255) return 1.f;
    int k = t - (e - 126);
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    return __uint_as_float((unsigned)(k + 127) << 23);
}

// v_cvt_pk_f16_f32 (round to nearest even), emitted by the COMPILER from the casts -- not inline assembly: an instruction that reads an
// MFMA result must wait the architected number of cycles after the MFMA, which the compiler's hazard recogniser inserts for instructions it
// knows and cannot insert for the operands of an asm block (found the hard way: with the conversions in asm, accumulator-fed splits read
// half-written registers and results differed by 2e-5 from run to run).
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
// a - float(h.lo) / a - float(h.hi): v_fma_mix_f32 reads an fp16 half as an operand of an fp32 fma -- ONE instruction, exact (the remainder
// of a rounding is representable).  The compiler does not form it from C (it emits v_cvt_f32_f16 + v_sub_f32), hence asm; that is safe
// here because h is the result of a compiler-emitted conversion OF a: by the time h exists, a has been readable for a while.
__device__ __forceinline__ float minus_lo_h(float a, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}
__device__ __forceinline__ float minus_hi_h(float a, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}

struct FmtB3 {                      // three bf16 pieces, six products: range of fp32, no scales
    static constexpr int NP = 3;
    static constexpr bool SCALED = false;
    using Op = X3;
    static __device__ __forceinline__ Op split(const f32x4 a, const f32x4 b) { return split8(a, b); }
    static __device__ __forceinline__ Op split_scaled(const f32x4 a, const f32x4 b, float, float) { return split8(a, b); }
    static __device__ __forceinline__ f32x4 mm(const Op& A, const Op& B, f32x4 c) { return mma6(A, B, c); }
    static __device__ __forceinline__ void put(u32x4* tab, int frag, int lane, const float (&v)[8], float) { put_frag(tab, frag, lane, v); }
    static __device__ __forceinline__ Op get(const u32x4* tab, int frag, int lo) { return get_frag(tab, frag, lo); }
};

struct FmtH2 {                      // two fp16 pieces, three products: operands scaled per class (see above)
    static constexpr int NP = 2;
    static constexpr bool SCALED = true;
    using Op = X2;
    // level by level over the four pairs, as split8 (a dependent vector instruction issues 8 cycles after its producer):
    // 4 conversions (half rate), 8 v_fma_mix_f32, 4 conversions = 26 issue cycles per pair against 41 for the bf16 x 3 split
    static __device__ __forceinline__ Op split(const f32x4 a, const f32x4 b) {
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        unsigned h[4], l[4];
        float r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = pk_f16(v[2 * i], v[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[2 * i] = minus_lo_h(v[2 * i], h[i]); r[2 * i + 1] = minus_hi_h(v[2 * i + 1], h[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = pk_f16(r[2 * i], r[2 * i + 1]);
        Op o;
        o.h = u32x4{h[0], h[1], h[2], h[3]};
        o.l = u32x4{l[0], l[1], l[2], l[3]};
        return o;
    }
    // slots 0..3 = a * sa, slots 4..7 = b * sb (powers of two: exact).  Table fills only (once per workgroup), so the scale is a plain
    // multiplication; v_fma_mixlo/hi_f16 would fold it into the conversion at the same issue cost (tools/probes/f16_split_rates.hip).
    static __device__ __forceinline__ Op split_scaled(const f32x4 a, const f32x4 b, float sa, float sb) {
        return split(a * sa, b * sb);
    }
    static __device__ __forceinline__ f32x4 mm1(const u32x4 a, const u32x4 b, const f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mm(const Op& A, const Op& B, f32x4 c) {      // smallest terms first
        c = mm1(A.l, B.h, c);
        c = mm1(A.h, B.l, c);
        c = mm1(A.h, B.h, c);
        return c;
    }
    static __device__ __forceinline__ void put(u32x4* tab, int frag, int lane, const float (&v)[8], float s) {
        const Op o = split_scaled(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, s, s);
        tab[(frag * 2 + 0) * 64 + lane] = o.h;
        tab[(frag * 2 + 1) * 64 + lane] = o.l;
    }
    static __device__ __forceinline__ Op get(const u32x4* tab, int frag, int lo) {
        Op r;
        r.h = tab[(frag * 2 + 0) * 64 + lo];
        r.l = tab[(frag * 2 + 1) * 64 + lo];
        return r;
    }
};

// max |p[i]|, i < n, over the workgroup (every thread gets it); scratch: one float per wave of the workgroup, not otherwise in use.
// Ends with a barrier: scratch may be reused right away.
__device__ __forceinline__ float block_absmax(const float* __restrict__ p, int n, float* scratch, int threads) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += threads) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < threads / 64; ++w) r = fmaxf(r, scratch[w]);
    __syncthreads();
    return r;
}

// the launch's gradient maximum from the slots its producer filled (stc_spmm_sum_f32 amax / the caller): max over n floats, wave-uniform
__device__ __forceinline__ float slots_max(const float* __restrict__ slots, int n) {
    float m = 0.f;
    for (int i = threadIdx.x & 63; i < n; i += 64) m = fmaxf(m, slots[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m)));      // (the builtin takes an int: pass the BITS, not the value)
}

// ---- activation operands of the fp16 x 2 format.  The reference's einsum / tanh are scale-free (STC_GNN.py:37-42, 72-78); two fp16 pieces
// are not: below 2^-3 the low piece is a subnormal (absolute floor 2^-25) and nothing exists above 65504.  Hence activations carry powers of
// two as well (exact), taken from maxima the kernels find themselves:
//   forward   ONE scale per node, from the maximum over every row the wave has loaded for that node (all planes; R*H <= H rides on it):
//             a wave reduction on the vector pipe (DPP) + scalar exponent arithmetic per node; the node's epilogue takes it out again.
//             The wave also keeps running per-plane maxima and leaves them in the launch's slots (one atomic max per plane and wave);
//   backward  the dW products sum over nodes in one accumulator, so they take ONE scale per plane and launch, from those slots
//             (the dW tiles of a plane are their own accumulators: the combine takes each plane's scale out).
// Non-negative floats order as their bit patterns: maxima are taken on the bits where that saves a conversion.
constexpr int STC_ACT_SLOTS = 256;          // slots per plane row of an activation-maximum buffer
// Where the maxima go.  Two fp16 pieces carry 22 bits for values of at least 2^-3, so a maximum in [2^(t-1), 2^t) leaves t + 2 binades below it at
// full precision -- what the planes of one node may differ by (S.X against X with graph row sums of 50: 5.6 binades).  The ceiling is set by the
// second-stage operands: a projected value is a sum of K x 32 products with a table normalised below 1, at most 96 x 2^8 < 65504 in the forward.
constexpr int STC_ACT_TARGET_FWD = 5;       // forward: the node's maximum over its rows into [2^4, 2^5)
constexpr int STC_ACT_TARGET_BWD = 6;       // backward: a plane's maximum into [2^5, 2^6); times at most 2^8 (RunScale) stays below 2^14
// Tables (W, T_c; normalised per workgroup from their own maxima): the same window argument -- Gaussian weights and the entries of
// T_2 = 2 Gc^2 - I spread over many binades, and with a maximum in [1/2, 1) (round 3) everything below an eighth of it sat on the absolute
// floor: with graph row sums of 50 that alone was 25x the reference's own fp32 noise in the forward (tests/test_scale_sweep.py).
//   forward   |U_c| <= 32 K x 2^5 x 2^4 = 49 152 at K = 3: the projected values still fit fp16 as operands of the category mix;
//   backward  |T_c dY| <= 32 x 2^2 x 2^(4+4): gradient fragments (node maximum in [2^3, 2^4), RunScale::ROOM_GRAD) against the mix table.
// W's block c = 0 carries sT as well (the two halves of a contraction meet in one accumulator): sT <= 2^11 keeps it inside fp16.
constexpr int STC_W_TARGET = 4;             // weight tables: maximum into [2^3, 2^4)
constexpr int STC_T_TARGET_FWD = 4;         // category-mix tables, forward
constexpr int STC_T_TARGET_BWD = 2;         // category-mix tables, backward
__device__ __forceinline__ float clamp_mix_scale(float sT) { return fminf(fmaxf(sT, 0.0625f), 2048.f); }

// (Vector-by-scalar products of the scaled kernels are written on the vector types and compile to v_pk_mul_f32: element by element -- twice the
// instructions -- the one-launch cell backward ran 1 270 instead of 1 212 us per launch and the gates forward 704 instead of 682; in these
// issue-bound loops the instruction count outweighs the packed instruction's extra cycles beside matrix instructions.)
// (Written in assembly as v_max3_f32 with |.| operand modifiers these maxima take a third of the instructions -- fmaxf gets a canonicalising
// v_max_f32 v, v, v per operand -- and the one-launch cell backward ran 1 214 instead of 1 163 us: the scheduler cannot move matrix
// instructions across asm statements.  Hence the plain form.)
__device__ __forceinline__ float absmax4(const f32x4 v) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
}
__device__ __forceinline__ float max_nonneg(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float vmax3_acc(float acc, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(acc, b), c); }      // (b, c non-negative)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float m) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(m), CTRL, 0xF, 0xF, true);
    return __builtin_fmaxf(m, __int_as_float(o));
}
// max of a NON-NEGATIVE value over the wave, wave-uniform (scalar register): quad swaps, half-row and row mirrors, then the four rows
__device__ __forceinline__ float wave_max_nonneg(float m) {
    m = dpp_max<0xB1>(m);            // quad_perm:[1,0,3,2]
    m = dpp_max<0x4E>(m);            // quad_perm:[2,3,0,1]
    m = dpp_max<0x141>(m);           // row_half_mirror
    m = dpp_max<0x140>(m);           // row_mirror: every lane of a row of 16 holds the row's maximum
    const int b = __float_as_int(m);
    const int r0 = __builtin_amdgcn_readlane(b, 0), r1 = __builtin_amdgcn_readlane(b, 16), r2 = __builtin_amdgcn_readlane(b, 32), r3 = __builtin_amdgcn_readlane(b, 48);
    const int a = r0 > r1 ? r0 : r1, c = r2 > r3 ? r2 : r3;
    return __int_as_float(a > c ? a : c);
}
__device__ __forceinline__ int wave_max_bits(float m) { return __float_as_int(wave_max_nonneg(m)); }
// a * b for two powers of two, on the exponent fields: integer arithmetic, i.e. the scalar unit when both are wave-uniform -- gfx950 has no
// scalar float multiply, and as v_mul_f32 every such product is a vector instruction AND a vector register that stays live through the node
// loop (seven spilled plane addresses in the order-3 gates backward: 27 % of its time).  The result must stay inside the normal range.
__device__ __forceinline__ float pow2_mul(float a, float b) { return __uint_as_float(__float_as_uint(a) + __float_as_uint(b) - 0x3F800000u); }
// a wave-uniform value into a scalar register (the compiler cannot know that a value read from LDS is the same in every lane)
__device__ __forceinline__ float uniform_bits(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// 1 / s for s = 2^k, |k| <= 126, on the exponent field
__device__ __forceinline__ float inv_pow2(float s) { return __uint_as_float(0x7F000000u - __float_as_uint(s)); }
// one wave's running maximum into its slot of a row of STC_ACT_SLOTS (the row was zero-filled before the launch)
__device__ __forceinline__ void leave_max(float* __restrict__ row, int slot, float m) {
    m = wave_max_nonneg(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(row) + (slot & (STC_ACT_SLOTS - 1)), __float_as_int(m));
}

// GRADIENT operands of the fp16 x 2 format.  What the gate prologues form (dHnew (Cand - H) U (1 - U), dRH H R (1 - R), dHnew U (1 - Cand^2)) can
// sit many binades below the state gradient (with states of 1e-8 round 3's launch-wide scale, taken from max |dHnew|, flushed them to zero),
// differ by tens of binades from node to node where gates saturate, and nothing is known about them before the launch.  So every NODE's
// gradient fragments get their own power of two a_n = 2^kn, from the node's own maximum (a wave reduction; target [2^3, 2^4)): the dZ tiles,
// which belong to the node, are unscaled by 1 / a_n on the spot.  The dW / db accumulators sum over the wave's nodes and carry the wave's
// REFERENCE scale a = 2^k, set by the first non-zero gradient the wave meets: a node joins them with its activation operand multiplied by
// a / a_n = 2^j -- the product of the two operands then has the accumulators' scale.
//   j < 0             the node's gradients are small against the reference: its activation operand shrinks and loses low bits exactly in
//                     proportion to how little the node adds to the sum;
//   0 < j <= 8 + 4    the node exceeds the reference: the excess goes to the activation operand (plane maximum in [2^5, 2^6): 2^8 of room)
//                     and, beyond that, into a_n itself (2^4: the fragment's second-stage operands T_c dY -- sums of 32 products with a table
//                     below 2^2 -- must fit fp16 too);
//   j > 12            the node cannot join at the sums' scale (its operands would leave fp16's range): the wave's PASS ends at this node.  The
//                     node is computed to its end at its own scale with its contributions to the sums muted (shift = 2^-24: RunScale::mute) and nothing -- or what the planes held -- stored, so the loop body has no exit of its own: the wave leaves the node
//                     loop at the latch and waits in the end-of-kernel combine, where what the workgroup's waves hold goes to the workgroup's
//                     partial row (every wave's own 1 / a applied).  While any wave of the workgroup has nodes left the kernel body runs again
//                     (tables refilled, operands requested afresh, sums empty; later passes ADD to the row): the stopped wave resumes AT its node,
//                     which now sets the reference.  (Rounds 3-4 zeroed the accumulators instead and lost up to half of the restarting node's own
//                     magnitude: node maxima 1, 64, 4096, 8192 dropped the 4096 node.)  Nothing touches the ~100 accumulator registers inside
//                     the loop (a rescaling multiplication there, even in a never-taken branch, put 10 - 50 scratch accesses per node into these
//                     one-wave-per-SIMD loops) and no wave polls another's state (an LDS flag read at every node top cost 130 instructions and
//                     4 - 12 scratch accesses per node through worse register allocation).  The price is paid where it happens: a stopped wave
//                     idles until the others finish their pass, so a launch with such jumps can take twice as long.  References only move
//                     towards larger gradients, by at least 2^13 per pass: at most ~16 passes whatever the data.  Jumps of that size between the
//                     nodes of one wave occur where gates saturate (factors e^-x: then by tens of binades).
// A node whose gradients are all zero joins at any scale and leaves the reference alone.  The wave's final 1 / a goes into the combine.  All
// bookkeeping is integer arithmetic on exponents (scalar unit); a non-finite maximum gives the smallest scale (the NaN goes where it has to).
__device__ __forceinline__ float exp2i(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }
struct RunScale {
    static constexpr int ROOM_ACT = 8, ROOM_GRAD = 4, EMPTY = 120;
    int k = EMPTY;                      // accumulators carry 2^k; EMPTY: they hold no gradient yet (the first non-zero node sets the reference)
    // mbits: the node's gradient maximum (bits of a non-negative float, wave-uniform) -> a_n; shift = a / a_n for the activation side and the
    // db sums; stop: the node is too large for the sums' scale -- the caller stores nothing of it (or what the planes held), ends its pass and
    // comes back to the node with empty sums.  The returned scale is then the node's own (nothing overflows on the way) and shift = mute() =
    // 2^-24 instead of the 2^j >= 2^13 the node would have needed: what it adds to the sums it cannot join is 2^-37 of its own contribution
    // (which the next pass adds in full) -- far below the sums' rounding -- without a select in the loop.  (Plane scales are powers of two
    // within 2^+-100, so the product of the two stays a normal number for pow2_mul.)
    static __device__ __forceinline__ float mute() { return exp2i(-24); }
    __device__ __forceinline__ float node(int mbits, float& shift, bool& stop) {
        int kn = 130 - ((mbits >> 23) & 255);                             // max 2^kn in [2^3, 2^4)
        kn = kn < -100 ? -100 : (kn > 100 ? 100 : kn);
        shift = 1.f;
        stop = false;
        if ((mbits & 0x7FFFFFFF) == 0) return 1.f;                        // no gradient at this node: zeros at any scale
        int j = k - kn;
        if (j > ROOM_ACT + ROOM_GRAD) {
            if (k != EMPTY) { stop = true; shift = mute(); return exp2i(kn); }
            k = kn; j = 0;
        }
        const int up = j > ROOM_ACT ? j - ROOM_ACT : 0;
        j -= up;
        shift = exp2i(j < -100 ? -100 : j);
        return exp2i(kn + up);
    }
    __device__ __forceinline__ float unscale() const { return exp2i(k == EMPTY ? 0 : -k); }      // 1 / a for the combine (sums of zeros: 1)
};

// A kernel's argument block read afresh from the kernarg segment, through a pointer the optimiser cannot see through.  The fp16 x 2 backward
// kernels run their body in passes (RunScale): with the arguments as ordinary kernel parameters everything the table fill and the combine
// need (a dozen pointers and sizes) stays live in scalar registers ACROSS the node loop of every pass, where there are none to spare -- the
// order-3 gates backward spilled seven plane addresses and reloaded them on every node.  Read at the top of a pass, they die where the
// single-pass kernel let them die.  T must be the kernel's ONLY parameter.
template <class T>
__device__ __forceinline__ T kernargs_fresh() {
#if defined(__HIP_DEVICE_COMPILE__)
    auto p = (const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
#else
    return T{};                                   // (the host pass of hipcc only parses device code)
#endif
}

// scale 2^k of the plane whose maxima sit in row `row` of the forward launch's slots (1 without slots: the caller gave no range information)
__device__ __forceinline__ float plane_scale(const float* __restrict__ zmax, int row) {
    if (!zmax) return 1.f;
    float m = 0.f;
    for (int i = threadIdx.x & 63; i < STC_ACT_SLOTS; i += 64) m = fmaxf(m, zmax[row * STC_ACT_SLOTS + i]);
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pow2_scale(wave_max_nonneg(m), STC_ACT_TARGET_BWD))));      // (a scalar register)
}

// row of a pair of 16-row tiles that slot (g, e) of an accumulator-fed operand stands for
__host__ __device__ constexpr int pair_row(int g, int e) { return 16 * (e >> 2) + 4 * g + (e & 3); }

#define kZero4 (f32x4{0.f, 0.f, 0.f, 0.f})

// Gate nonlinearities of the fused epilogues: stc_common.h (hardware exp2 / rcp, accurate relative to the result for every argument)
__device__ __forceinline__ float fast_sigmoid(float v) { return stc_sigmoid(v); }
__device__ __forceinline__ float fast_tanh(float v) { return stc_tanh(v); }

}  // namespace
