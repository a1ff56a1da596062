// Two-ring patch aggregation: the gradient of a state from its pieces AND the transpose aggregation of the candidate's gradient that
// follows it, in one launch (reference STC_GNN.py:65-79 through autograd, planar cells; the two launches it replaces: stc_spmm_sum_f32
// with its blend epilogue, then the plain stc_patch_spmm_f32 of dY):
//
//     dH  = sum_k D_k + S.(A [+ A2])          the state's gradient: direct planes + the aggregation of the aggregated planes
//     dY  = dH * U * (1 - Cand^2)             the candidate pre-activation's gradient (GRU blend + tanh backward) -- NEVER written
//     dBm = S.dY                              what the post-aggregation candidate backward needs beside dY (dY itself it re-forms)
//
// A written plane costs the metric step twice what a read one does: dY existed only to cross the launch boundary
// between the two aggregations.  Here a workgroup owns a PATCH of <= 32 interior rows (the tiles of stc_spmm_patch.hip), forms dH and dY
// for the patch's FIRST RING (the interior and every row the interior's S-rows touch: <= 64 rows, 60 on a 4 x 8 tile of the 8-neighbour
// grid) out of the SECOND ring's rows of A staged in LDS (<= 96 rows), parks dY in LDS and aggregates it for the interior.  The first
// ring's halo is computed redundantly (1.9 x the gathers; its addends come from L2, where the neighbouring patches leave them).
//
// Column chunks of 32 float4 (512 bytes of a row): half a wave per row, 96 x 512 B = 48 KiB of LDS + 6 KiB of tables, two workgroups per
// compute unit.  Per chunk: staged registers -> LDS | next chunk requested | first ring: gather, addends, dY -> registers | dY -> LDS
// (over the staged rows) | interior: gather, store.  Tables ((LDS offset, value) pairs) sit in LDS and are read as broadcasts.
#include "stc_common.h"

#include <atomic>

namespace {

constexpr int R2_THREADS = 256, R2_WAVES = R2_THREADS / 64;
constexpr int R2_INT = STC_RING2_INTERIOR;       // interior rows per patch (32)
constexpr int R2_L1 = STC_RING2_FIRST;           // first-ring slots, interior included (64)
constexpr int R2_L2 = STC_RING2_SECOND;          // second-ring rows staged (96)
constexpr int R2_W = STC_RING2_WIDTH;            // table entries per row (8)
constexpr int R2_Q = 32;                         // float4 pieces per chunk and row: half a wave
constexpr int R2_STAGE = R2_L2 / (2 * R2_WAVES); // rows a half-wave stages per chunk (12)
constexpr int R2_S1 = R2_L1 / (2 * R2_WAVES);    // first-ring slots per half-wave (8)
constexpr int R2_S2 = R2_INT / (2 * R2_WAVES);   // interior rows per half-wave (4)
constexpr int R2_MAX_ADD = 5;

using v4f = __attribute__((ext_vector_type(4))) float;

struct Ring2Plan {
    const int32_t *l2_rows, *l1_rows, *int_rows;       // (P, 96) staged rows; (P, 64) first-ring rows (-1: empty; bit 30: interior); (P, 32) interior rows (-1: empty)
    const int32_t *t1, *t2;                            // (P, 64, 8, 2) / (P, 32, 8, 2): (LDS byte offset of the source row, value bits)
    int n_patches;
};

struct Ring2Args {
    Ring2Plan pl;
    const v4f *A, *A2;                                 // gathered operands (B, n, F4)
    const v4f* add[R2_MAX_ADD];
    int n_add;
    const v4f *U, *Cand;                               // BLEND form: Cand = the previous state H
    v4f *Y, *Y2, *Z;                                   // SUM: dH, -, dBm;  BLEND: Cand, Hnew, S.Hnew;  CHAIN: V (or null), -, Z
    int n, F4;
    // CHAIN: scales of the two aggregations and the INTERIOR's addends (the first ring's are add[])
    float alpha1, alpha2;
    const v4f* add0[R2_MAX_ADD];
    float scale0[R2_MAX_ADD];
};

// What the first ring computes from the gathered sum and the slot's own operands, and which of it the second aggregation takes:
//   R2_SUM    dH = sum + addends (stored for interior rows);            V = dH U (1 - Cand^2)              -> Z = S.V = dBm
//   R2_BLEND  Cand = tanh(sum + A), Hnew = (1 - U) H + U Cand (both stored: the forward of stc_spmm_blend_fwd_f32);  V = Hnew  -> Z = S.Hnew
//   R2_CHAIN  V = alpha1 sum + addends (stored for interior rows if Y is given)   -> Z = alpha2 S.V + sum_k scale0[k] add0[k]: two chained
//             aggregations of the order-3 feature recurrence -- forward 2 S.(S.X) - X (STC_GNN.py:24-29 applied to the feature side, :37), and
//             its transpose in Clenshaw form, d0 - d2 + S^T (d1 + 2 S^T d2)
enum { R2_SUM = 0, R2_BLEND = 1, R2_CHAIN = 2 };

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE, bool HAS_A2, int NADD, int N0 = 0>                 // NADD: addends of the first ring;  N0: of the interior (CHAIN)
__global__ __launch_bounds__(R2_THREADS, 2) void ring2_sum_kernel(Ring2Args a) {
    static_assert(MODE != R2_BLEND || (NADD == 1 && !HAS_A2), "the blend form: one addend (A), one gathered operand (Bm)");
    static_assert((MODE == R2_CHAIN) == (N0 > 0), "interior addends: the chain form, at least one");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    v4f* tile = reinterpret_cast<v4f*>(lds);                                           // [96][32] staged rows, then [64][32] dY
    int2* tab1 = reinterpret_cast<int2*>(lds + (size_t)R2_L2 * R2_Q * 16);             // [64][8]
    int2* tab2 = tab1 + R2_L1 * R2_W;                                                  // [32][8]

    const int p = stc_xcd_tile(blockIdx.x, a.pl.n_patches);
    if (p < 0) return;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, q = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = wave * 2 + half;                                                    // half-wave 0..7: rows / slots hw, hw + 8, ...
    const int n_chunks = a.F4 / R2_Q;

    for (int e = tid; e < R2_L1 * R2_W; e += R2_THREADS) tab1[e] = reinterpret_cast<const int2*>(a.pl.t1)[(size_t)p * R2_L1 * R2_W + e];
    for (int e = tid; e < R2_INT * R2_W; e += R2_THREADS) tab2[e] = reinterpret_cast<const int2*>(a.pl.t2)[(size_t)p * R2_INT * R2_W + e];

    int stage_row[R2_STAGE], l1_row[R2_S1], in_row[R2_S2];
#pragma unroll
    for (int k = 0; k < R2_STAGE; ++k) stage_row[k] = a.pl.l2_rows[(size_t)p * R2_L2 + k * 8 + hw];
#pragma unroll
    for (int i = 0; i < R2_S1; ++i) l1_row[i] = a.pl.l1_rows[(size_t)p * R2_L1 + i * 8 + hw];
#pragma unroll
    for (int i = 0; i < R2_S2; ++i) in_row[i] = a.pl.int_rows[(size_t)p * R2_INT + i * 8 + hw];

    // 32-bit piece offsets inside a plane (the entry point checks batch * n * F4 < 2^28): one register per row instead of a 64-bit address per
    // row AND plane -- with size_t indices the kernel held 64 address registers and spilled
    const unsigned base = (unsigned)b * (unsigned)a.n * (unsigned)a.F4 + (unsigned)q;
    unsigned stage_off[R2_STAGE], l1_off[R2_S1];
#pragma unroll
    for (int k = 0; k < R2_STAGE; ++k) stage_off[k] = base + (unsigned)stage_row[k] * (unsigned)a.F4;
#pragma unroll
    for (int i = 0; i < R2_S1; ++i) l1_off[i] = base + (unsigned)(l1_row[i] >= 0 ? (l1_row[i] & 0x3FFFFFFF) : 0) * (unsigned)a.F4;
    v4f st[R2_STAGE];
    auto request = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < R2_STAGE; ++k) {
            const unsigned at = stage_off[k] + chunk * R2_Q;
            st[k] = a.A[at];
            if (HAS_A2) st[k] += a.A2[at];
        }
    };
    // The first ring's own operands (addends, U, Cand: NADD + 2 pieces per slot) are requested ONE SLOT AHEAD, and the next chunk's staged rows
    // only after the last slot's request: memory returns in issue order, so a slot's request made behind the staging request waits for 12 rows
    // from HBM (the first version: one exposed latency per slot; holding all eight slots' operands a chunk ahead spilled 120 registers).
    constexpr int NOP = MODE == R2_CHAIN ? NADD : NADD + 2, NOPA = NOP > 0 ? NOP : 1;
    auto slot_at = [&](int i, int chunk) { return l1_off[i] + (unsigned)(chunk * R2_Q); };
    auto request_slot = [&](v4f (&o)[NOPA], int i, int chunk) {
        const unsigned at = slot_at(i, chunk);
#pragma unroll
        for (int k = 0; k < NADD; ++k) o[k] = a.add[k][at];
        if constexpr (MODE != R2_CHAIN) {
            o[NADD] = a.U[at];
            o[NADD + 1] = a.Cand[at];
        }
    };
    v4f opn[NOPA];
    request(0);
    request_slot(opn, 0, 0);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        if (chunk) lds_only_barrier();                                                 // the previous chunk's interior sums are done with the tile
#pragma unroll
        for (int k = 0; k < R2_STAGE; ++k) tile[(k * 8 + hw) * R2_Q + q] = st[k];
        lds_only_barrier();                                                            // (the first chunk: the tables are in place too)
        const bool more = chunk + 1 < n_chunks;
        // ---- first ring: dH = addends + S.A out of the tile; dY = dH U (1 - Cand^2)
        v4f dy[R2_S1];
#pragma unroll
        for (int i = 0; i < R2_S1; ++i) {
            const int slot = i * 8 + hw, row = l1_row[i];
            const bool interior = row >= 0 && (row & (1 << 30));
            v4f cur[NOPA];
#pragma unroll
            for (int k = 0; k < NOP; ++k) cur[k] = opn[k];
            if (i + 1 < R2_S1) {
                request_slot(opn, i + 1, chunk);
            } else if (more) {                                                         // the next chunk: its staged rows, then its first slot
                request(chunk + 1);
                request_slot(opn, 0, chunk + 1);
            }
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < R2_W; ++w) {
                const int2 t = tab1[slot * R2_W + w];
                const v4f x = *reinterpret_cast<const v4f*>(lds + t.x + q * 16);
                const float v = __int_as_float(t.y);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, x[c], acc[c]);
            }
            v4f dh = MODE == R2_CHAIN ? a.alpha1 * acc : acc;
#pragma unroll
            for (int k = 0; k < NADD; ++k) dh += cur[k];
            if constexpr (MODE == R2_CHAIN) {
                if (a.Y != nullptr && interior) __builtin_nontemporal_store(dh, a.Y + slot_at(i, chunk));
                dy[i] = dh;
            } else if constexpr (MODE == R2_SUM) {
                const v4f u = cur[NOPA - 2], cd = cur[NOPA - 1];
                if (interior) __builtin_nontemporal_store(dh, a.Y + slot_at(i, chunk));
#pragma unroll
                for (int c = 0; c < 4; ++c) dy[i][c] = dh[c] * u[c] * (1.f - cd[c] * cd[c]);
            } else {                                                                   // cd = the previous state H
                const v4f u = cur[NOPA - 2], cd = cur[NOPA - 1];
                v4f cand;
#pragma unroll
                for (int c = 0; c < 4; ++c) { cand[c] = stc_tanh(dh[c]); dy[i][c] = (1.f - u[c]) * cd[c] + u[c] * cand[c]; }
                if (interior) {
                    __builtin_nontemporal_store(cand, a.Y + slot_at(i, chunk));
                    __builtin_nontemporal_store(dy[i], a.Y2 + slot_at(i, chunk));
                }
            }
        }
        lds_only_barrier();                                                            // every wave is done with the staged rows
#pragma unroll
        for (int i = 0; i < R2_S1; ++i) tile[(i * 8 + hw) * R2_Q + q] = dy[i];
        // CHAIN: the interior rows' own addends, requested now that the first ring's registers are free (they return behind the next chunk's staged
        // rows, which the next tile write waits for in any case)
        v4f own[R2_S2][N0 > 0 ? N0 : 1];
        if constexpr (MODE == R2_CHAIN) {
#pragma unroll
            for (int i = 0; i < R2_S2; ++i)
#pragma unroll
                for (int k = 0; k < N0; ++k)
                    own[i][k] = a.add0[k][base + (unsigned)(in_row[i] >= 0 ? in_row[i] : 0) * (unsigned)a.F4 + (unsigned)(chunk * R2_Q)];
        }
        lds_only_barrier();
        // ---- interior: dBm = S.dY out of the tile
#pragma unroll
        for (int i = 0; i < R2_S2; ++i) {
            const int r = i * 8 + hw, row = in_row[i];
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < R2_W; ++w) {
                const int2 t = tab2[r * R2_W + w];
                const v4f x = *reinterpret_cast<const v4f*>(lds + t.x + q * 16);
                const float v = __int_as_float(t.y);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, x[c], acc[c]);
            }
            if constexpr (MODE == R2_CHAIN) {
                acc *= a.alpha2;
#pragma unroll
                for (int k = 0; k < N0; ++k) acc += a.scale0[k] * own[i][k];
            }
            if (row >= 0) __builtin_nontemporal_store(acc, a.Z + (base + (unsigned)row * (unsigned)a.F4 + (unsigned)(chunk * R2_Q)));
        }
    }
}

template <int MODE, bool HAS_A2>
int launch_ring2(const Ring2Args& a, int batch, hipStream_t s, int n_add0 = 0) {
    const size_t lds = (size_t)R2_L2 * R2_Q * 16 + (size_t)(R2_L1 + R2_INT) * R2_W * sizeof(int2);
    static_assert((size_t)R2_L2 * R2_Q * 16 + (size_t)(R2_L1 + R2_INT) * R2_W * sizeof(int2) <= 64 * 1024, "within the default dynamic-LDS limit: no attribute to set");
    using Kernel = void (*)(Ring2Args);
    Kernel kern = nullptr;
    if constexpr (MODE == R2_BLEND) {
        kern = ring2_sum_kernel<R2_BLEND, false, 1>;
    } else if constexpr (MODE == R2_CHAIN) {
        static constexpr Kernel table[3][R2_MAX_ADD] = {
#define STC_R2_ROW(NA) {ring2_sum_kernel<R2_CHAIN, HAS_A2, NA, 1>, ring2_sum_kernel<R2_CHAIN, HAS_A2, NA, 2>, ring2_sum_kernel<R2_CHAIN, HAS_A2, NA, 3>, \
                        ring2_sum_kernel<R2_CHAIN, HAS_A2, NA, 4>, ring2_sum_kernel<R2_CHAIN, HAS_A2, NA, 5>}
            STC_R2_ROW(0), STC_R2_ROW(1), STC_R2_ROW(2)};
#undef STC_R2_ROW
        kern = table[a.n_add][n_add0 - 1];
    } else {
        switch (a.n_add) {
            case 0: kern = ring2_sum_kernel<R2_SUM, HAS_A2, 0>; break;
            case 1: kern = ring2_sum_kernel<R2_SUM, HAS_A2, 1>; break;
            case 2: kern = ring2_sum_kernel<R2_SUM, HAS_A2, 2>; break;
            case 3: kern = ring2_sum_kernel<R2_SUM, HAS_A2, 3>; break;
            case 4: kern = ring2_sum_kernel<R2_SUM, HAS_A2, 4>; break;
            default: kern = ring2_sum_kernel<R2_SUM, HAS_A2, 5>; break;
        }
    }
    const int per = (a.pl.n_patches + stc::kNumXcd - 1) / stc::kNumXcd;
    hipLaunchKernelGGL(kern, dim3(per * stc::kNumXcd, batch), dim3(R2_THREADS), lds, s, a);
    STC_LAUNCH_CHECK("stc_ring2 launch");
    return STC_OK;
}

int check_ring2(const char* who, const void* l2_rows, const void* l1_rows, const void* int_rows, const void* t1, const void* t2, int n_patches, int n_rows,
                int batch, int C, int h) {
    STC_REQUIRE(h == 16 && C >= 1 && (C * h) % (4 * R2_Q) == 0, STC_EUNSUPPORTED, "%s: rows of C * h = %d floats (hidden 16, whole 512-byte chunks)", who, C * h);
    STC_REQUIRE(batch >= 0 && batch <= 65535 && n_rows >= 0 && n_patches >= 0, STC_EINVAL, "%s: bad sizes", who);
    STC_REQUIRE((long long)batch * n_rows * (C * h / 4) < (1ll << 28), STC_ELIMIT, "%s: planes of %lld 16-byte pieces (32-bit offsets: < 2^28)", who,
                (long long)batch * n_rows * (C * h / 4));
    if (batch == 0 || n_rows == 0) return STC_OK;
    STC_REQUIRE(n_patches >= 1 && (long long)n_patches * R2_INT >= n_rows, STC_EINVAL, "%s: %d patches cannot cover %d rows", who, n_patches, n_rows);
    STC_REQUIRE(l2_rows && l1_rows && int_rows && t1 && t2, STC_EINVAL, "%s: null plan array", who);
    STC_REQUIRE(reinterpret_cast<uintptr_t>(t1) % 8 == 0 && reinterpret_cast<uintptr_t>(t2) % 8 == 0, STC_EALIGN, "%s: tables must be 8-byte aligned", who);
    return STC_OK;
}

}  // namespace

extern "C" int stc_ring2_sum_f32(const int32_t* l2_rows, const int32_t* l1_rows, const int32_t* int_rows, const int32_t* t1, const int32_t* t2,
                                 int32_t n_patches, int32_t n_rows,
                                 const float* A, const float* A2, int32_t n_add, const float* const* add,
                                 const float* U, const float* Cand, float* Y, float* Z,
                                 int32_t batch, int32_t C, int32_t h, void* stream) {
    STC_REQUIRE(n_add >= 0 && n_add <= R2_MAX_ADD, STC_ELIMIT, "stc_ring2_sum_f32: 0..%d addends, got %d", R2_MAX_ADD, n_add);
    if (int rc = check_ring2("stc_ring2_sum_f32", l2_rows, l1_rows, int_rows, t1, t2, n_patches, n_rows, batch, C, h)) return rc;
    if (batch == 0 || n_rows == 0) return STC_OK;
    STC_REQUIRE(A && U && Cand && Y && Z && (n_add == 0 || add), STC_EINVAL, "stc_ring2_sum_f32: null pointer");
    STC_REQUIRE(stc::aligned16(A) && (!A2 || stc::aligned16(A2)) && stc::aligned16(U) && stc::aligned16(Cand) && stc::aligned16(Y) && stc::aligned16(Z),
                STC_EALIGN, "stc_ring2_sum_f32: planes must be 16-byte aligned");
    STC_REQUIRE(Y != A && Y != A2 && Z != A && Z != A2 && Y != Z && Y != U && Y != Cand && Z != U && Z != Cand, STC_EINVAL,
                "stc_ring2_sum_f32: results must not alias the gathered operands, the first ring's gate planes or each other");
    Ring2Args a{};
    a.pl = Ring2Plan{l2_rows, l1_rows, int_rows, t1, t2, n_patches};
    a.A = reinterpret_cast<const v4f*>(A);
    a.A2 = reinterpret_cast<const v4f*>(A2);
    a.n_add = n_add;
    for (int i = 0; i < n_add; ++i) {
        STC_REQUIRE(add[i] && stc::aligned16(add[i]) && add[i] != Y && add[i] != Z, STC_EINVAL, "stc_ring2_sum_f32: addend %d null, misaligned or aliasing a result", i);
        a.add[i] = reinterpret_cast<const v4f*>(add[i]);
    }
    a.U = reinterpret_cast<const v4f*>(U);
    a.Cand = reinterpret_cast<const v4f*>(Cand);
    a.Y = reinterpret_cast<v4f*>(Y);
    a.Z = reinterpret_cast<v4f*>(Z);
    a.n = n_rows;
    a.F4 = C * h / 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return A2 ? launch_ring2<R2_SUM, true>(a, batch, s) : launch_ring2<R2_SUM, false>(a, batch, s);
}

extern "C" int stc_ring2_blend_f32(const int32_t* l2_rows, const int32_t* l1_rows, const int32_t* int_rows, const int32_t* t1, const int32_t* t2,
                                   int32_t n_patches, int32_t n_rows,
                                   const float* Bm, const float* A, const float* U, const float* H,
                                   float* Cand, float* Hnew, float* SHnew,
                                   int32_t batch, int32_t C, int32_t h, void* stream) {
    if (int rc = check_ring2("stc_ring2_blend_f32", l2_rows, l1_rows, int_rows, t1, t2, n_patches, n_rows, batch, C, h)) return rc;
    if (batch == 0 || n_rows == 0) return STC_OK;
    STC_REQUIRE(Bm && A && U && H && Cand && Hnew && SHnew, STC_EINVAL, "stc_ring2_blend_f32: null pointer");
    for (const void* q : {(const void*)Bm, (const void*)A, (const void*)U, (const void*)H, (const void*)Cand, (const void*)Hnew, (const void*)SHnew})
        STC_REQUIRE(stc::aligned16(q), STC_EALIGN, "stc_ring2_blend_f32: planes must be 16-byte aligned");
    for (const float* out : {Cand, Hnew, SHnew})
        STC_REQUIRE(out != Bm && out != A && out != U && out != H, STC_EINVAL, "stc_ring2_blend_f32: a result aliases an operand (the first ring re-reads them)");
    STC_REQUIRE(Cand != Hnew && Cand != SHnew && Hnew != SHnew, STC_EINVAL, "stc_ring2_blend_f32: results alias each other");
    Ring2Args a{};
    a.pl = Ring2Plan{l2_rows, l1_rows, int_rows, t1, t2, n_patches};
    a.A = reinterpret_cast<const v4f*>(Bm);
    a.n_add = 1;
    a.add[0] = reinterpret_cast<const v4f*>(A);
    a.U = reinterpret_cast<const v4f*>(U);
    a.Cand = reinterpret_cast<const v4f*>(H);
    a.Y = reinterpret_cast<v4f*>(Cand);
    a.Y2 = reinterpret_cast<v4f*>(Hnew);
    a.Z = reinterpret_cast<v4f*>(SHnew);
    a.n = n_rows;
    a.F4 = C * h / 4;
    return launch_ring2<R2_BLEND, false>(a, batch, static_cast<hipStream_t>(stream));
}

extern "C" int stc_ring2_chain_f32(const int32_t* l2_rows, const int32_t* l1_rows, const int32_t* int_rows, const int32_t* t1, const int32_t* t2,
                                   int32_t n_patches, int32_t n_rows,
                                   const float* A, const float* A2, float alpha1, int32_t n_add1, const float* const* add1, float* V,
                                   float alpha2, int32_t n_add0, const float* const* add0, const float* scale0, float* Z,
                                   int32_t batch, int32_t C, int32_t h, void* stream) {
    STC_REQUIRE(n_add1 >= 0 && n_add1 <= 2, STC_ELIMIT, "stc_ring2_chain_f32: 0..2 first-ring addends, got %d", n_add1);
    STC_REQUIRE(n_add0 >= 1 && n_add0 <= R2_MAX_ADD, STC_ELIMIT, "stc_ring2_chain_f32: 1..%d interior addends, got %d", R2_MAX_ADD, n_add0);
    if (int rc = check_ring2("stc_ring2_chain_f32", l2_rows, l1_rows, int_rows, t1, t2, n_patches, n_rows, batch, C, h)) return rc;
    if (batch == 0 || n_rows == 0) return STC_OK;
    STC_REQUIRE(A && Z && add0 && (n_add1 == 0 || add1), STC_EINVAL, "stc_ring2_chain_f32: null pointer");
    STC_REQUIRE(stc::aligned16(A) && (!A2 || stc::aligned16(A2)) && (!V || stc::aligned16(V)) && stc::aligned16(Z), STC_EALIGN,
                "stc_ring2_chain_f32: planes must be 16-byte aligned");
    STC_REQUIRE(Z != A && Z != A2 && V != A && (!V || V != A2) && V != Z, STC_EINVAL, "stc_ring2_chain_f32: results must not alias the gathered operands or each other");
    Ring2Args a{};
    a.pl = Ring2Plan{l2_rows, l1_rows, int_rows, t1, t2, n_patches};
    a.A = reinterpret_cast<const v4f*>(A);
    a.A2 = reinterpret_cast<const v4f*>(A2);
    a.n_add = n_add1;
    for (int i = 0; i < n_add1; ++i) {
        STC_REQUIRE(add1[i] && stc::aligned16(add1[i]) && add1[i] != Z && add1[i] != V, STC_EINVAL, "stc_ring2_chain_f32: first-ring addend %d null, misaligned or aliasing a result", i);
        a.add[i] = reinterpret_cast<const v4f*>(add1[i]);
    }
    for (int i = 0; i < n_add0; ++i) {
        STC_REQUIRE(add0[i] && stc::aligned16(add0[i]) && add0[i] != Z && add0[i] != V, STC_EINVAL, "stc_ring2_chain_f32: interior addend %d null, misaligned or aliasing a result", i);
        a.add0[i] = reinterpret_cast<const v4f*>(add0[i]);
        a.scale0[i] = scale0 ? scale0[i] : 1.f;
    }
    a.alpha1 = alpha1;
    a.alpha2 = alpha2;
    a.Y = reinterpret_cast<v4f*>(V);
    a.Z = reinterpret_cast<v4f*>(Z);
    a.n = n_rows;
    a.F4 = C * h / 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return A2 ? launch_ring2<R2_CHAIN, true>(a, batch, s, n_add0) : launch_ring2<R2_CHAIN, false>(a, batch, s, n_add0);
}
