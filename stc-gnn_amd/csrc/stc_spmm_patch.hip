// LDS-staged "patch" form of the spatial (1-mode) aggregation  Y = alpha S.X + beta Y0  on gfx950 (reference STC_GNN.py:37,
// torch.einsum('bncl,nm->bmcl', X, T_n(Gs)): the same product as stc_bcsr_spmm_f32, csrc/stc_spmm.hip).
//
// The row-blocked kernel gathers every neighbour row of a block of 4 output rows through L1 / L2: 4.5 row fetches per output row
// on the 8-neighbour grid, all but one of them cache hits -- and it is those hits (L2 -> CU traffic, not HBM) that hold it at
// 0.61 of the HBM peak.  Here the host clusters the graph's rows into PATCHES (graph.py _patch_plan: up to 32 output rows whose
// neighbour rows number at most 64 together; 1.85 source rows per output row on that grid as 4 x 8 tiles) and one workgroup per (patch, batch
// element) copies the patch's source rows into LDS once per column chunk -- each lane a 16-byte piece, a wave instruction one
// whole 1 KiB chunk of a row -- and forms the 32 output rows out of LDS:
//
//   chunk c:   LDS <- registers (requested during chunk c-2)  |  request chunk c+2 into registers  |  32 rows x 64 lanes: sum over the
//              row's entries of val x LDS[source position], non-temporal store
//
// so that a workgroup has its next two chunks in flight while it multiplies (tools/probes/spmm_patch_sweep.hip: 180 us for the
// bench's unit against 211 us row-blocked and 181 us for a plain copy of the same two planes).  The sum of a row runs over its
// entries in CSR order with one fmaf each, from zero: bit for bit the row-blocked and the CSR kernels' result.
#include "stc_common.h"

#include <atomic>
#include <type_traits>

namespace {

constexpr int PT_THREADS = 256;
constexpr int PT_WAVES = PT_THREADS / 64;
constexpr int PT_ROWS = STC_PATCH_ROWS;          // output rows per patch
constexpr int PT_SRC = STC_PATCH_MAX_SRC;        // source rows per patch (LDS: PT_SRC KiB per 256-float chunk)
constexpr int PT_Q = 64;                         // float4 pieces per chunk: one per lane
constexpr int PT_PER = PT_SRC / PT_WAVES;        // source rows each wave requests per chunk

using v4f = __attribute__((ext_vector_type(4))) float;

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

// A 16-byte piece of a feature row and its fp32 sums: four floats, or eight bf16 (bf16 storage, BASELINE configuration 5: values widen to
// fp32 exactly, products and sums in fp32, ONE rounding to bf16 at the store -- the contract of csrc/stc_spmm_bf16.hip, same fmaf chain).
template <bool BF16>
struct Piece;

template <>
struct Piece<false> {                                     // four floats (kept as ONE vector value: element arrays cost this kernel 16 registers)
    v4f v;
    __device__ __forceinline__ void zero() { v = v4f{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void fma(float s, const v4f x) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaf(s, x[c], v[c]);
    }
    template <bool HAS_Y0>                                // alpha * sum (+ beta * y0), as the piece to store
    __device__ __forceinline__ v4f finish(float alpha, float beta, const v4f y0) const {
        v4f out;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = HAS_Y0 ? fmaf(beta, y0[c], alpha * v[c]) : alpha * v[c];
        return out;
    }
};

template <>
struct Piece<true> {                                      // eight bf16 columns, summed in fp32
    float v[8];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    __device__ __forceinline__ void fma(float s, const v4f x) {
        const u32x4 u = __builtin_bit_cast(u32x4, x);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = fmaf(s, __uint_as_float(u[i] << 16), v[2 * i]);
            v[2 * i + 1] = fmaf(s, __uint_as_float(u[i] & 0xffff0000u), v[2 * i + 1]);
        }
    }
    template <bool HAS_Y0>
    __device__ __forceinline__ v4f finish(float alpha, float beta, const v4f y0) const {
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = alpha * v[i];
        if (HAS_Y0) {
            const u32x4 u = __builtin_bit_cast(u32x4, y0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[2 * i] = fmaf(beta, __uint_as_float(u[i] << 16), r[2 * i]);
                r[2 * i + 1] = fmaf(beta, __uint_as_float(u[i] & 0xffff0000u), r[2 * i + 1]);
            }
        }
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x2 t = {(__bf16)r[2 * i], (__bf16)r[2 * i + 1]};           // v_cvt_pk_bf16_f32, round to nearest even
            o[i] = __builtin_bit_cast(unsigned, t);
        }
        return __builtin_bit_cast(v4f, o);
    }
};

struct PatchPlan {
    const int32_t *src, *rows, *cnt;
    const uint8_t* idx;
    const float* val;
    int n_patches, width;
};

// Workgroup barrier that orders LDS only.  __syncthreads() carries a workgroup-scope fence that drains vmcnt as well: every chunk would
// wait for the previous chunk's result stores to be acknowledged -- the latency this kernel is built to keep off its critical path.  The
// tile is the only memory the waves of a workgroup share; global results are written once and read by no one in the launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ v4f patch_dump[64];                           // where the slots of a patch that hold no row store (never read for a result)

// W = entries per row (the plan's width, compile time: the row loop is unrolled over it).  Each wave owns the patch's rows wave,
// wave + 4, ...: RPW = 8 of them, whose RPW x W (LDS offset, value) pairs sit one per lane in NV registers each and reach the
// scalar unit by v_readlane with a constant lane number -- no table reads (and their latency) between the tile reads of a row.
//
// One workgroup per (patch, batch element), dispatched in order: XCD x (workgroup id % 8, the hardware's round-robin) works through the
// x-th eighth of the patch list front to back, about 64 patches in flight, and a patch's neighbours in the list find the rows they share
// in that XCD's L2 (the host lists each eighth row by row, graph.py _grid_tiles / _patch_plan).  (A persistent variant -- two workgroups per compute
// unit walking strided runs of patches, tables loaded once per patch -- hid the per-workgroup table loads but lost that: a workgroup's
// next patch was 64 further down the list, the rows shared with it long evicted; FETCH_SIZE 1.45 x the matrix against 1.05 x.)
template <int W, bool HAS_Y0, bool BF16>
__global__ __launch_bounds__(PT_THREADS, 2) void spmm_patch_kernel(PatchPlan pl, int n_rows, int n_cols, const v4f* __restrict__ X,
                                                               const v4f* Y0, v4f* Y, int F4, float alpha, float beta) {
    // (Y0 may BE Y -- the backward sums run in place, include/stc_hip.h -- and a ragged patch stores a row twice through its duplicated slots:
    //  neither pointer may carry __restrict__, or the compiler may sink a Y0 load below the store of the row's duplicate.)
    constexpr int RPW = PT_ROWS / PT_WAVES, NV = (RPW * W + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    v4f* halo = reinterpret_cast<v4f*>(lds_raw);                                   // [PT_SRC][PT_Q]

    const int p = stc_xcd_tile(blockIdx.x, pl.n_patches);
    if (p < 0) return;                                   // whole workgroup leaves together
    const int b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int n_chunks = F4 / PT_Q;

    // source rows first (the chunk requests wait for them): the patch's list is a fixed record of PT_SRC row numbers laid out wave by
    // wave (wave w stages positions w, w + 4, ...: its 16 numbers are contiguous), padded with repeats of its first source row -- one
    // scalar load per wave, straight into the registers the request's base addresses are formed from; no branches in the staging.
    int my_src[PT_PER];
    {
        const int32_t* rec = pl.src + ((size_t)p * PT_WAVES + wave) * PT_PER;
#pragma unroll
        for (int k = 0; k < PT_PER; ++k) my_src[k] = rec[k];
    }
    // the wave's table: entry e = i * W + w belongs to its i-th row (patch row wave + 4 i), lane e % 64 of register e / 64
    int t_off[NV];
    float t_val[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int e = v * 64 + lane, i = e / W, w = e % W;
        const size_t at = ((size_t)p * PT_ROWS + (i < RPW ? wave + PT_WAVES * i : 0)) * W + w;
        t_off[v] = i < RPW ? (int)pl.idx[at] * (PT_Q * 16) : 0;
        t_val[v] = i < RPW ? pl.val[at] : 0.f;
    }
    // rows and entry counts of the wave's rows: lane i
    const int v_row = lane < RPW ? pl.rows[(size_t)p * PT_ROWS + wave + PT_WAVES * lane] : -1;
    const int v_cnt = lane < RPW ? pl.cnt[(size_t)p * PT_ROWS + wave + PT_WAVES * lane] : 0;

    const v4f* Xb = X + (size_t)b * n_cols * F4 + lane;
    // TWO chunks in flight per workgroup (register sets A and B, 16 pieces per lane each): with one, a workgroup had nothing outstanding
    // while it waited for its first chunk and little while it summed -- 2 x 60 KiB per compute unit kept the chip at 5.2 TB/s.
    v4f nxa[PT_PER], nxb[PT_PER];
    auto request = [&](v4f (&nx)[PT_PER], int chunk) {
#pragma unroll
        for (int k = 0; k < PT_PER; ++k) nx[k] = Xb[(size_t)my_src[k] * F4 + chunk * PT_Q];
    };
    const unsigned char* tile = lds_raw + lane * 16;
    // One chunk: registers -> tile, the next chunk requested, the wave's rows summed out of the tile.  No branch in it: the wait for the
    // NEXT staging can then be counted past this chunk's result stores (vmcnt(8): they stay in flight), where behind a branch the
    // compiler has to drain them -- one exposed store latency per chunk.
    // (With a Y0 operand ONE chunk ahead: memory returns in issue order, so the Y0 pieces of a chunk must be asked for before any later
    //  chunk's rows or their wait is a wait for those as well -- and a second set of Y0 registers does not fit beside two chunks.)
    // (... and with tables of more than 8 entries per row the second set does not fit in the 256 registers two workgroups per compute unit leave)
    constexpr int DEPTH = (HAS_Y0 || W > 8) ? 1 : 2;
    auto step = [&](v4f (&nx)[PT_PER], v4f (&other)[PT_PER], int chunk, auto more) {
        if (chunk) lds_barrier();                         // the previous chunk's sums are done with the tile
#pragma unroll
        for (int k = 0; k < PT_PER; ++k) halo[(k * PT_WAVES + wave) * PT_Q + lane] = nx[k];
        lds_barrier();
        // Y0 pieces of the wave's rows BEFORE the next request (memory returns in issue order: asked for after it, their wait would also
        // be a wait for the whole next chunk)
        v4f y0[HAS_Y0 ? RPW : 1];
        if (HAS_Y0) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int row = __builtin_amdgcn_readlane(v_row, i);
                y0[i] = __builtin_nontemporal_load(row < 0 ? patch_dump + lane : Y0 + ((size_t)b * n_rows + row) * F4 + chunk * PT_Q + lane);
            }
        }
        if (decltype(more)::value) {
            if (DEPTH == 2) request(nx, chunk + 2);      // into the set just emptied
            else request(other, chunk + 1);
        }
        // (a slot of the patch without a row -- patches of 1 .. 3 rows only, graph.py -- computes like the others and stores into a dump line)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = __builtin_amdgcn_readlane(v_row, i);
            const size_t o = ((size_t)b * n_rows + (row < 0 ? 0 : row)) * F4 + chunk * PT_Q + lane;
            Piece<BF16> acc;
            acc.zero();
            // all W entries (the host fills a row's tail with zero-weight repeats of its last entry: the same sum, term for term)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int e = i * W + w;
                const int off = __builtin_amdgcn_readlane(t_off[e / 64], e % 64);
                const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_val[e / 64]), e % 64));
                acc.fma(v, *reinterpret_cast<const v4f*>(tile + off));
            }
            if (__builtin_amdgcn_readlane(v_cnt, i) == 0) acc.zero();                          // a row without entries sums nothing
            __builtin_nontemporal_store(acc.template finish<HAS_Y0>(alpha, beta, y0[HAS_Y0 ? i : 0]), row < 0 ? patch_dump + lane : Y + o);
        }
    };
    constexpr std::true_type more{};
    constexpr std::false_type last{};
    request(nxa, 0);
    int chunk = 0;
    if (DEPTH == 2) {
        if (n_chunks > 1) request(nxb, 1);
        for (; chunk + 3 < n_chunks; chunk += 2) {        // pairs of chunks with both successors to request: no branch inside
            step(nxa, nxb, chunk, more);
            step(nxb, nxa, chunk + 1, more);
        }
        const int rest = n_chunks - chunk;                // 1, 2 or 3 chunks left
        if (rest == 3) {
            step(nxa, nxb, chunk, more);
            step(nxb, nxa, chunk + 1, last);
            step(nxa, nxb, chunk + 2, last);
        } else if (rest == 2) {
            step(nxa, nxb, chunk, last);
            step(nxb, nxa, chunk + 1, last);
        } else {
            step(nxa, nxb, chunk, last);
        }
    } else {
        for (; chunk + 2 < n_chunks; chunk += 2) {
            step(nxa, nxb, chunk, more);
            step(nxb, nxa, chunk + 1, more);
        }
        if (n_chunks - chunk == 2) {
            step(nxa, nxb, chunk, more);
            step(nxb, nxa, chunk + 1, last);
        } else {
            step(nxa, nxb, chunk, last);
        }
    }
}

}  // namespace

namespace {

// X / Y0 / Y as 16-byte pieces: F4 of them per row (F / 4 floats or F / 8 bf16)
int launch_patch(const char* who, const PatchPlan& pl, int n_rows, int n_cols, const void* X, const void* Y0, void* Y, int batch, int F4, bool bf16,
                 float alpha, float beta, hipStream_t s) {
    const size_t lds = (size_t)PT_SRC * PT_Q * 16;
    const bool has_y0 = Y0 != nullptr && beta != 0.f;
    using Kernel = void (*)(PatchPlan, int, int, const v4f*, const v4f*, v4f*, int, float, float);
    Kernel kern = nullptr;
    int slot = 0;
#define STC_PATCH_W(W_, SLOT_) case W_: kern = bf16 ? (has_y0 ? spmm_patch_kernel<W_, true, true> : spmm_patch_kernel<W_, false, true>)  \
                                                    : (has_y0 ? spmm_patch_kernel<W_, true, false> : spmm_patch_kernel<W_, false, false>); slot = SLOT_; break
    switch (pl.width) { STC_PATCH_W(4, 0); STC_PATCH_W(8, 1); STC_PATCH_W(12, 2); STC_PATCH_W(16, 3); STC_PATCH_W(24, 4); STC_PATCH_W(32, 5);
                        default: STC_REQUIRE(false, STC_EUNSUPPORTED, "%s: width %d (built for 4, 8, 12, 16, 24, 32)", who, pl.width); }
#undef STC_PATCH_W
    static std::atomic<int> granted[6][2][2][16];        // 64 KiB of dynamic LDS is above the default limit: once per kernel and device
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;      // a device the table does not cover: granted on every launch
    if (dev < 0 || !granted[slot][has_y0][bf16][dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return stc::hip_status(e, who);
        if (dev >= 0) granted[slot][has_y0][bf16][dev].store(1, std::memory_order_release);
    }
    const int per = (pl.n_patches + stc::kNumXcd - 1) / stc::kNumXcd;
    hipLaunchKernelGGL(kern, dim3(per * stc::kNumXcd, batch), dim3(PT_THREADS), lds, s, pl, n_rows, n_cols, static_cast<const v4f*>(X),
                       static_cast<const v4f*>(Y0), static_cast<v4f*>(Y), F4, alpha, beta);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return stc::hip_status(e, who);
    return STC_OK;
}

int check_patch(const char* who, const void* patch_src, const void* patch_rows, const void* patch_cnt, const void* patch_idx, const void* patch_val,
                int n_patches, int n_rows, int n_cols, const void* X, const void* Y0, const void* Y, int batch, int F, int chunk_elems, float beta) {
    STC_REQUIRE(patch_src && patch_rows && patch_cnt && patch_idx && patch_val && X && Y, STC_EINVAL, "%s: null pointer", who);
    STC_REQUIRE(n_patches >= 1 && (long long)n_patches * PT_ROWS >= n_rows, STC_EINVAL, "%s: %d patches of %d rows cannot cover %d rows", who, n_patches, PT_ROWS, n_rows);
    STC_REQUIRE(F % chunk_elems == 0, STC_EUNSUPPORTED, "%s: F=%d must be a multiple of %d (rows in whole 1 KiB chunks)", who, F, chunk_elems);
    STC_REQUIRE(beta == 0.f || Y0, STC_EINVAL, "%s: beta != 0 needs Y0", who);
    STC_REQUIRE(X != Y, STC_EINVAL, "%s: X must not alias Y", who);
    STC_REQUIRE(stc::aligned16(X) && stc::aligned16(Y) && (!Y0 || stc::aligned16(Y0)), STC_EALIGN, "%s: X / Y / Y0 must be 16-byte aligned", who);
    STC_REQUIRE(batch <= 65535, STC_ELIMIT, "%s: batch %d > 65535 (grid.y)", who, batch);
    return STC_OK;
}

}  // namespace

extern "C" int stc_patch_spmm_f32(const int32_t* patch_src, const int32_t* patch_rows, const int32_t* patch_cnt,
                                  const uint8_t* patch_idx, const float* patch_val, int32_t n_patches, int32_t width,
                                  int32_t n_rows, int32_t n_cols, const float* X, const float* Y0, float* Y,
                                  int32_t batch, int32_t F, float alpha, float beta, void* stream) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && F >= 0 && n_patches >= 0, STC_EINVAL, "stc_patch_spmm_f32: negative size");
    if (n_rows == 0 || batch == 0 || F == 0) return STC_OK;
    if (int rc = check_patch("stc_patch_spmm_f32", patch_src, patch_rows, patch_cnt, patch_idx, patch_val, n_patches, n_rows, n_cols, X, Y0, Y, batch, F, 4 * PT_Q, beta)) return rc;
    const PatchPlan pl{patch_src, patch_rows, patch_cnt, patch_idx, patch_val, n_patches, width};
    return launch_patch("stc_patch_spmm_f32 launch", pl, n_rows, n_cols, X, Y0, Y, batch, F / 4, false, alpha, beta, static_cast<hipStream_t>(stream));
}

extern "C" int stc_patch_spmm_bf16(const int32_t* patch_src, const int32_t* patch_rows, const int32_t* patch_cnt,
                                   const uint8_t* patch_idx, const float* patch_val, int32_t n_patches, int32_t width,
                                   int32_t n_rows, int32_t n_cols, const void* X, const void* Y0, void* Y,
                                   int32_t batch, int32_t F, float alpha, float beta, void* stream) {
    STC_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && F >= 0 && n_patches >= 0, STC_EINVAL, "stc_patch_spmm_bf16: negative size");
    if (n_rows == 0 || batch == 0 || F == 0) return STC_OK;
    if (int rc = check_patch("stc_patch_spmm_bf16", patch_src, patch_rows, patch_cnt, patch_idx, patch_val, n_patches, n_rows, n_cols, X, Y0, Y, batch, F, 8 * PT_Q, beta)) return rc;
    const PatchPlan pl{patch_src, patch_rows, patch_cnt, patch_idx, patch_val, n_patches, width};
    return launch_patch("stc_patch_spmm_bf16 launch", pl, n_rows, n_cols, X, Y0, Y, batch, F / 8, true, alpha, beta, static_cast<hipStream_t>(stream));
}
