// Node kernel of the BDG_Dif layer: category (2-mode) product + the K*K*L concat
// + projection + bias, forward and backward, any (C, L, Ho, Ks, Kc) -- gfx950.
//
// Reference span: STC_GNN.py:38-45 (forward) and its autograd.  Per node row r
// (one (batch, node) pair; C category rows of L features in each of the Ks
// Chebyshev slabs Z_n):
//
//   U[c', (c,o)]  = sum_{n,l} Z_n[r,c',l] * W[(n,c,l), o]            "project"   (rows x KD) . (KD x J)
//   Y[d, o]       = bias[o] + U[d,(0,o)] + sum_{c>=1} sum_{c'} T_c[c',d] * U[c',(c,o)]   "mix"
//
// i.e. project first, mix categories second: the concat of the reference is
// never formed, T_0 = I costs nothing, and the mix runs on Ho (<= L) columns.
// KD = Ks*L, J = Kc*Ho.  Backward, with Q[c',(c,o)] = sum_d T_c[c',d] dY[d,o]:
//
//   dZ_n[c',l] = sum_{c,o} Q[c',(c,o)] W[(n,c,l),o]       dW[(n,c,l),o] = sum_rows Z_n[c',l] Q[c',(c,o)]
//   db[o] = sum_rows dY[d,o]                               dT_c[c',d] = sum_{node,o} U[c',(c,o)] dY[d,o]
//
// This file is the general fp32 VALU version (LDS-tiled, persistent workgroups);
// weight-gradient partials are summed in a fixed order by a second kernel, so
// results are bitwise reproducible.
#include "stc_common.h"
#include "stc_node_mfma.h"

#include <atomic>
#include <cstdlib>

namespace {

constexpr int NODE_THREADS = 256;
constexpr int NODE_BWD_MAX_GRID = 512;

struct ZPtrs { const float* p[STC_MAX_K]; };
struct DZPtrs { float* p[STC_MAX_K]; };

struct NodeDims {
    int Ks, Kc, C, L, Lw, Ho;   // Lw <= L: feature rows of W per block; slab columns [Lw, L) are padding
    int KD, J, KDp, Jp;     // contraction widths and their round-ups to 4
    int TN, rows;           // nodes per tile, category rows per tile (TN*C)
    int zs, qs, ds, wts;    // LDS row strides (floats): Zt, Ut/Qt, dYt, WsT
};

inline int up4(int v) { return (v + 3) & ~3; }

NodeDims make_dims(int Ks, int Kc, int C, int L, int Lw, int Ho) {
    NodeDims d;
    d.Ks = Ks; d.Kc = Kc; d.C = C; d.L = L; d.Lw = Lw; d.Ho = Ho;
    d.KD = Ks * L; d.J = Kc * Ho;
    d.KDp = up4(d.KD); d.Jp = up4(d.J);
    d.TN = C >= 32 ? 1 : 32 / C;
    d.rows = d.TN * C;
    d.zs = d.KDp + 4; d.qs = d.Jp + 4; d.ds = up4(Ho) + 4; d.wts = d.KDp + 4;
    return d;
}

// forward LDS carve (floats)
struct FwdCarve { int Ws, Ts, bias, Zt, Ut, total; };
FwdCarve fwd_carve(const NodeDims& d) {
    FwdCarve c; int o = 0;
    c.Ws = o;   o += d.KD * d.Jp;
    c.Ts = o;   o += up4((d.Kc > 1 ? d.Kc - 1 : 0) * d.C * d.C);
    c.bias = o; o += up4(d.Ho);
    c.Zt = o;   o += d.rows * d.zs;
    c.Ut = o;   o += d.rows * d.qs;
    c.total = o;
    return c;
}

// backward LDS carve (floats)
struct BwdCarve { int WsT, Ts, Zt, dYt, Qt, Ut, dWacc, dbacc, dTacc, total; };
BwdCarve bwd_carve(const NodeDims& d, bool want_dT) {
    BwdCarve c; int o = 0;
    const int nT = (d.Kc > 1 ? d.Kc - 1 : 0) * d.C * d.C;
    c.WsT = o;   o += d.J * d.wts;
    c.Ts = o;    o += up4(nT);
    c.Zt = o;    o += d.rows * d.zs;
    c.dYt = o;   o += d.rows * d.ds;
    c.Qt = o;    o += d.rows * d.qs;
    c.Ut = o;    o += want_dT ? d.rows * d.qs : 0;
    c.dWacc = o; o += d.KDp * d.Jp;
    c.dbacc = o; o += up4(d.Ho);
    c.dTacc = o; o += want_dT ? up4(nT) : 0;
    c.total = o;
    return c;
}

__device__ __forceinline__ void fma4(float4& a, float s, const float4& w) {
    a.x = fmaf(s, w.x, a.x); a.y = fmaf(s, w.y, a.y); a.z = fmaf(s, w.z, a.z); a.w = fmaf(s, w.w, a.w);
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(NODE_THREADS) void bdg_node_fwd_kernel(
    ZPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, long long total_rows, NodeDims d, FwdCarve cv, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem + cv.Ws;
    float* Ts = smem + cv.Ts;
    float* sb = smem + cv.bias;
    float* Zt = smem + cv.Zt;
    float* Ut = smem + cv.Ut;
    const int tid = threadIdx.x;
    const int CC = d.C * d.C;

    // stage W as Ws[k=(n,l)][j=(c,o)], T_1.. and the bias once per workgroup
    for (int idx = tid; idx < d.KD * d.Jp; idx += NODE_THREADS) {
        const int k = idx / d.Jp, j = idx - k * d.Jp;
        float v = 0.f;
        if (j < d.J) {
            const int n = k / d.L, l = k - n * d.L, c = j / d.Ho, o = j - c * d.Ho;
            if (l < d.Lw) v = W[((size_t)(n * d.Kc + c) * d.Lw + l) * d.Ho + o];
        }
        Ws[idx] = v;
    }
    for (int idx = tid; idx < (d.Kc - 1) * CC; idx += NODE_THREADS) Ts[idx] = Tc[CC + idx];
    for (int idx = tid; idx < d.Ho; idx += NODE_THREADS) sb[idx] = bias ? bias[idx] : 0.f;
    __syncthreads();

    const int n_cg = d.Jp / 4;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = (long long)tile * d.rows;
        const int rows_here = (int)min((long long)d.rows, total_rows - row0);
        for (int n = 0; n < d.Ks; ++n) {
            const float* src = Z.p[n] + row0 * d.L;
            for (int idx = tid; idx < d.rows * d.L; idx += NODE_THREADS) {
                const int r = idx / d.L, l = idx - r * d.L;
                Zt[r * d.zs + n * d.L + l] = idx < rows_here * d.L ? src[idx] : 0.f;
            }
        }
        __syncthreads();
        // project: Ut[r][j0..j0+3] = sum_k Zt[r][k] * Ws[k][j0..j0+3]
        for (int mt = tid; mt < d.rows * n_cg; mt += NODE_THREADS) {
            const int r = mt / n_cg, j0 = (mt - r * n_cg) * 4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* zr = Zt + r * d.zs;
            for (int k = 0; k < d.KD; ++k) fma4(acc, zr[k], *reinterpret_cast<const float4*>(Ws + k * d.Jp + j0));
            *reinterpret_cast<float4*>(Ut + r * d.qs + j0) = acc;
        }
        __syncthreads();
        // mix: Y[r=(node,dd)][o] = b[o] + U[r][(0,o)] + sum_{c>=1} sum_{c'} T_c[c'][dd] U[(node,c')][(c,o)]
        for (int e = tid; e < rows_here * d.Ho; e += NODE_THREADS) {
            const int r = e / d.Ho, o = e - r * d.Ho;
            const int node = r / d.C, dd = r - node * d.C;
            float y = sb[o] + Ut[r * d.qs + o];
            for (int c = 1; c < d.Kc; ++c) {
                const float* T = Ts + (c - 1) * CC + dd;
                const float* Uc = Ut + (node * d.C) * d.qs + c * d.Ho + o;
                for (int cp = 0; cp < d.C; ++cp) y = fmaf(T[cp * d.C], Uc[cp * d.qs], y);
            }
            Y[(row0 + r) * d.Ho + o] = y;
        }
        // no barrier needed here: the next tile's load only writes Zt, which the mix does not read,
        // and its project phase (which rewrites Ut) sits behind the barrier after that load.
    }
}

// ------------------------------------------------------------------ backward
__global__ __launch_bounds__(NODE_THREADS) void bdg_node_bwd_kernel(
    ZPtrs Z, const float* __restrict__ Tc, const float* __restrict__ W, const float* __restrict__ dY,
    DZPtrs dZ, float* __restrict__ partial, long long total_rows, NodeDims d, BwdCarve cv,
    int n_tiles, int want_dT, int want_db) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* WsT = smem + cv.WsT;
    float* Ts = smem + cv.Ts;
    float* Zt = smem + cv.Zt;
    float* dYt = smem + cv.dYt;
    float* Qt = smem + cv.Qt;
    float* Ut = smem + cv.Ut;
    float* dWacc = smem + cv.dWacc;
    float* dbacc = smem + cv.dbacc;
    float* dTacc = smem + cv.dTacc;
    const int tid = threadIdx.x;
    const int CC = d.C * d.C;
    const int nT = (d.Kc - 1) * CC;

    // WsT[j=(c,o)][k=(n,l)] (k padded with zeros), T_1.., zeroed accumulators and pad columns
    for (int idx = tid; idx < d.J * d.wts; idx += NODE_THREADS) {
        const int j = idx / d.wts, k = idx - j * d.wts;
        float v = 0.f;
        if (k < d.KD) {
            const int n = k / d.L, l = k - n * d.L, c = j / d.Ho, o = j - c * d.Ho;
            if (l < d.Lw) v = W[((size_t)(n * d.Kc + c) * d.Lw + l) * d.Ho + o];
        }
        WsT[idx] = v;
    }
    for (int idx = tid; idx < nT; idx += NODE_THREADS) Ts[idx] = Tc[CC + idx];
    for (int idx = tid; idx < d.KDp * d.Jp; idx += NODE_THREADS) dWacc[idx] = 0.f;
    for (int idx = tid; idx < d.Ho; idx += NODE_THREADS) dbacc[idx] = 0.f;
    if (want_dT) for (int idx = tid; idx < nT; idx += NODE_THREADS) dTacc[idx] = 0.f;
    for (int idx = tid; idx < d.rows * d.zs; idx += NODE_THREADS) Zt[idx] = 0.f;
    for (int idx = tid; idx < d.rows * d.qs; idx += NODE_THREADS) Qt[idx] = 0.f;
    __syncthreads();

    const int n_kg = d.KDp / 4, n_jg = d.Jp / 4;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = (long long)tile * d.rows;
        const int rows_here = (int)min((long long)d.rows, total_rows - row0);
        for (int n = 0; n < d.Ks; ++n) {
            const float* src = Z.p[n] + row0 * d.L;
            for (int idx = tid; idx < d.rows * d.L; idx += NODE_THREADS) {
                const int r = idx / d.L, l = idx - r * d.L;
                Zt[r * d.zs + n * d.L + l] = idx < rows_here * d.L ? src[idx] : 0.f;
            }
        }
        {
            const float* src = dY + row0 * d.Ho;
            for (int idx = tid; idx < d.rows * d.Ho; idx += NODE_THREADS) {
                const int r = idx / d.Ho, o = idx - r * d.Ho;
                dYt[r * d.ds + o] = idx < rows_here * d.Ho ? src[idx] : 0.f;
            }
        }
        __syncthreads();
        // Q[r=(node,c')][(c,o)] = sum_dd T_c[c'][dd] dY[(node,dd)][o]   (c = 0: Q = dY)
        for (int e = tid; e < d.rows * d.J; e += NODE_THREADS) {
            const int r = e / d.J, j = e - r * d.J;
            const int c = j / d.Ho, o = j - c * d.Ho;
            float q;
            if (c == 0) {
                q = dYt[r * d.ds + o];
            } else {
                const int node = r / d.C, cp = r - node * d.C;
                const float* T = Ts + (c - 1) * CC + cp * d.C;
                const float* g = dYt + (node * d.C) * d.ds + o;
                q = 0.f;
                for (int dd = 0; dd < d.C; ++dd) q = fmaf(T[dd], g[dd * d.ds], q);
            }
            Qt[r * d.qs + j] = q;
        }
        if (want_dT) {   // U recomputed for dT_c: U[r][j] = sum_k Zt[r][k] * WsT[j][k]
            for (int e = tid; e < d.rows * d.J; e += NODE_THREADS) {
                const int r = e / d.J, j = e - r * d.J;
                const float* zr = Zt + r * d.zs;
                const float* wr = WsT + j * d.wts;
                float u = 0.f;
                for (int k = 0; k < d.KD; ++k) u = fmaf(zr[k], wr[k], u);
                Ut[r * d.qs + j] = u;
            }
        }
        __syncthreads();
        // dZ[r][k0..k0+3] = sum_j Q[r][j] * WsT[j][k0..k0+3]
        for (int mt = tid; mt < rows_here * n_kg; mt += NODE_THREADS) {
            const int r = mt / n_kg, k0 = (mt - r * n_kg) * 4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* qr = Qt + r * d.qs;
            for (int j = 0; j < d.J; ++j) fma4(acc, qr[j], *reinterpret_cast<const float4*>(WsT + j * d.wts + k0));
            const float a[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + q;
                if (k < d.KD) {
                    const int n = k / d.L, l = k - n * d.L;
                    dZ.p[n][(row0 + r) * d.L + l] = a[q];
                }
            }
        }
        // dW[k0..k0+3][j0..j0+3] += sum_r Zt[r][k] * Q[r][j]     (4x4 block per thread, accumulators in LDS)
        for (int blk = tid; blk < n_kg * n_jg; blk += NODE_THREADS) {
            const int k0 = (blk / n_jg) * 4, j0 = (blk - (blk / n_jg) * n_jg) * 4;
            float4 a0 = *reinterpret_cast<float4*>(dWacc + (k0 + 0) * d.Jp + j0);
            float4 a1 = *reinterpret_cast<float4*>(dWacc + (k0 + 1) * d.Jp + j0);
            float4 a2 = *reinterpret_cast<float4*>(dWacc + (k0 + 2) * d.Jp + j0);
            float4 a3 = *reinterpret_cast<float4*>(dWacc + (k0 + 3) * d.Jp + j0);
            for (int r = 0; r < d.rows; ++r) {
                const float4 z = *reinterpret_cast<const float4*>(Zt + r * d.zs + k0);
                const float4 q = *reinterpret_cast<const float4*>(Qt + r * d.qs + j0);
                fma4(a0, z.x, q); fma4(a1, z.y, q); fma4(a2, z.z, q); fma4(a3, z.w, q);
            }
            *reinterpret_cast<float4*>(dWacc + (k0 + 0) * d.Jp + j0) = a0;
            *reinterpret_cast<float4*>(dWacc + (k0 + 1) * d.Jp + j0) = a1;
            *reinterpret_cast<float4*>(dWacc + (k0 + 2) * d.Jp + j0) = a2;
            *reinterpret_cast<float4*>(dWacc + (k0 + 3) * d.Jp + j0) = a3;
        }
        if (want_db) {
            for (int o = tid; o < d.Ho; o += NODE_THREADS) {
                float s = dbacc[o];
                for (int r = 0; r < d.rows; ++r) s += dYt[r * d.ds + o];
                dbacc[o] = s;
            }
        }
        if (want_dT) {   // dT_c[c'][dd] += sum_{node,o} U[(node,c')][(c,o)] * dY[(node,dd)][o]
            for (int e = tid; e < nT; e += NODE_THREADS) {
                const int c1 = e / CC, rem = e - c1 * CC;
                const int cp = rem / d.C, dd = rem - cp * d.C;
                float s = dTacc[e];
                for (int node = 0; node < d.TN; ++node) {
                    const float* u = Ut + (node * d.C + cp) * d.qs + (c1 + 1) * d.Ho;
                    const float* g = dYt + (node * d.C + dd) * d.ds;
                    for (int o = 0; o < d.Ho; ++o) s = fmaf(u[o], g[o], s);
                }
                dTacc[e] = s;
            }
        }
        __syncthreads();   // Zt/dYt/Qt/Ut are rewritten by the next tile
    }

    // this workgroup's partial sums, already in destination layout: [dW (Ks*Kc*L*Ho) | db (Ho) | dT (Kc*C*C)]
    const int nW = d.Ks * d.Kc * d.Lw * d.Ho;
    float* out = partial + (size_t)blockIdx.x * (nW + d.Ho + d.Kc * CC);
    for (int idx = tid; idx < nW; idx += NODE_THREADS) {
        const int o = idx % d.Ho;
        int t = idx / d.Ho;
        const int l = t % d.Lw; t /= d.Lw;
        const int c = t % d.Kc, n = t / d.Kc;
        out[idx] = dWacc[(n * d.L + l) * d.Jp + c * d.Ho + o];
    }
    for (int idx = tid; idx < d.Ho; idx += NODE_THREADS) out[nW + idx] = dbacc[idx];
    for (int idx = tid; idx < d.Kc * CC; idx += NODE_THREADS)
        out[nW + d.Ho + idx] = (want_dT && idx >= CC) ? dTacc[idx - CC] : 0.f;
}

// Fixed-order sum of the per-workgroup partials into dW | db | dT.  16 output elements per workgroup,
// 16 thread groups each summing every 16th partial row, then a fixed-order combine through LDS:
// bitwise reproducible, and wide enough (stride/16 workgroups) to take microseconds, not 100+.
constexpr int RED_ELEMS = 16, RED_GROUPS = NODE_THREADS / RED_ELEMS;
__global__ __launch_bounds__(NODE_THREADS) void bdg_node_reduce_kernel(
    const float* __restrict__ partial, int n_parts, int stride, int nW, int Ho, int nT,
    float* dW, float* db, float* dT) {
    __shared__ float red[RED_GROUPS][RED_ELEMS];
    const int le = threadIdx.x % RED_ELEMS, g = threadIdx.x / RED_ELEMS;
    const int e = blockIdx.x * RED_ELEMS + le;
    float s = 0.f;
    if (e < stride)
        for (int p = g; p < n_parts; p += RED_GROUPS) s += partial[(size_t)p * stride + e];
    red[g][le] = s;
    __syncthreads();
    if (g != 0 || e >= stride) return;
    for (int k = 1; k < RED_GROUPS; ++k) s += red[k][le];
    if (e < nW) dW[e] = s;
    else if (e < nW + Ho) { if (db) db[e - nW] = s; }
    else if (dT) dT[e - nW - Ho] = s;
}

int check_dims(const char* who, int Ks, int Kc, int C, int L, int Lw, int Ho, long long nodes) {
    STC_REQUIRE(Lw >= 1 && Lw <= L, STC_EINVAL, "%s: Lw=%d must be in [1, L=%d]", who, Lw, L);
    STC_REQUIRE(Ks >= 1 && Ks <= STC_MAX_K && Kc >= 1 && Kc <= STC_MAX_K, STC_ELIMIT,
                "%s: Chebyshev orders Ks=%d Kc=%d outside [1,%d]", who, Ks, Kc, STC_MAX_K);
    STC_REQUIRE(C >= 1 && L >= 1 && Ho >= 1 && nodes >= 0, STC_EINVAL,
                "%s: bad sizes C=%d L=%d Ho=%d nodes=%lld", who, C, L, Ho, nodes);
    STC_REQUIRE(nodes * (long long)C < (1ll << 31), STC_ELIMIT, "%s: nodes*C = %lld exceeds 2^31", who, nodes * (long long)C);
    return STC_OK;
}

// Dispatch ceiling of the node / cell kernels (stc_set_dispatch_level): 0 = every path (split-operand matrix cores, then fp32 MFMA, then the
// generic kernels), 1 = no split-operand kernels, 2 = generic kernels only.  A process-wide setting for tests and A/B runs -- the entry
// points themselves read no environment.
std::atomic<int> g_dispatch_level{0};
bool mfma_enabled() { return g_dispatch_level.load(std::memory_order_relaxed) < 2; }
bool x3_enabled() { return g_dispatch_level.load(std::memory_order_relaxed) < 1; }

int bwd_grid(const NodeDims& d, long long nodes) {
    const long long n_tiles = (nodes + d.TN - 1) / d.TN;
    return (int)(n_tiles < NODE_BWD_MAX_GRID ? n_tiles : NODE_BWD_MAX_GRID);
}

}  // namespace

// fixed-order sum of per-workgroup partial rows [dW | db] (stride nW + Ho) -- shared with stc_node_bf16.hip
int stc_node_reduce_partials(const float* partial, int n_parts, int nW, int Ho, float* dW, float* db, hipStream_t s) {
    const int stride = nW + Ho;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}

extern "C" int stc_bdg_node_fwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                                    const float* W, const float* bias, float* Y,
                                    int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    if (int rc = check_dims("stc_bdg_node_fwd_f32", Ks, Kc, C, L, Lw, Ho, nodes)) return rc;
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(Z && W && Y && (Kc == 1 || Tc), STC_EINVAL, "stc_bdg_node_fwd_f32: null Z/W/Y/Tc");
    ZPtrs zp{};
    for (int n = 0; n < Ks; ++n) {
        STC_REQUIRE(Z[n], STC_EINVAL, "stc_bdg_node_fwd_f32: Z[%d] is null", n);
        zp.p[n] = Z[n];
    }
    if (x3_enabled()) {
        const int rc = stc_node_fwd_x3(Z, Ks, Tc, Kc, W, bias, Y, nodes, C, L, Lw, Ho, static_cast<hipStream_t>(stream));
        if (rc != STC_NOT_HANDLED) return rc;
    }
    if (mfma_enabled()) {
        const int rc = stc_node_fwd_mfma(Z, Ks, Tc, Kc, W, bias, Y, nodes, C, L, Lw, Ho, static_cast<hipStream_t>(stream));
        if (rc != STC_NOT_HANDLED) return rc;
    }
    const NodeDims d = make_dims(Ks, Kc, C, L, Lw, Ho);
    const FwdCarve cv = fwd_carve(d);
    const size_t lds = (size_t)cv.total * sizeof(float);
    STC_REQUIRE(lds <= stc::kMaxLdsBytes, STC_ELIMIT,
                "stc_bdg_node_fwd_f32: needs %zu B of LDS (C=%d L=%d Ho=%d Ks=%d Kc=%d), limit %zu", lds, C, L, Ho, Ks, Kc, stc::kMaxLdsBytes);
    if (int rc = stc::hip_status(stc::allow_lds(bdg_node_fwd_kernel, lds), "hipFuncSetAttribute(node fwd)")) return rc;
    const long long n_tiles = (nodes + d.TN - 1) / d.TN;
    const int grid = (int)(n_tiles < 4 * stc::kNumCu ? n_tiles : 4 * stc::kNumCu);
    hipLaunchKernelGGL(bdg_node_fwd_kernel, dim3(grid), dim3(NODE_THREADS), lds, static_cast<hipStream_t>(stream),
                       zp, Tc, W, bias, Y, (long long)nodes * C, d, cv, (int)n_tiles);
    STC_LAUNCH_CHECK("stc_bdg_node_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_cell_fused_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t h) {
    return (mfma_enabled() && Ks == Kc && stc_cell_fused_shape_ok(Ks, C, L, h)) ? 1 : 0;
}

extern "C" int stc_cell_gates_fwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                                      const float* W, const float* bias, const float* H,
                                      float* U, float* Rg, float* CandIn,
                                      int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, int32_t cin, void* stream) {
    if (int rc = check_dims("stc_cell_gates_fwd_f32", Ks, Kc, C, L, Lw, 2 * h, nodes)) return rc;
    STC_REQUIRE(cin >= 0 && cin + h <= L, STC_EINVAL, "stc_cell_gates_fwd_f32: cin=%d + h=%d exceed the row width L=%d", cin, h, L);
    if (!stc_cell_fused_supported(Ks, Kc, C, L, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_fwd_f32: shape not on the fused path");
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(Z && W && H && U && Rg && CandIn && (Kc == 1 || Tc), STC_EINVAL, "stc_cell_gates_fwd_f32: null pointer");
    for (int n = 0; n < Ks; ++n) STC_REQUIRE(Z[n], STC_EINVAL, "stc_cell_gates_fwd_f32: Z[%d] is null", n);
    STC_REQUIRE(Z[0] != CandIn, STC_EINVAL, "stc_cell_gates_fwd_f32: CandIn must not alias Z[0]");
    int rc = STC_NOT_HANDLED;
    if (x3_enabled()) rc = stc_cell_gates_fwd_x3(Z, Ks, Tc, W, bias, H, U, Rg, CandIn, nodes, C, L, Lw, cin, static_cast<hipStream_t>(stream));
    if (rc == STC_NOT_HANDLED) rc = stc_cell_gates_fwd_mfma(Z, Ks, Tc, W, bias, H, U, Rg, CandIn, nodes, C, L, Lw, cin, static_cast<hipStream_t>(stream));
    return rc == STC_NOT_HANDLED ? stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_fwd_f32: operands not usable by the fused path (alignment)") : rc;
}

extern "C" int stc_cell_gates_bwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc, const float* W,
                                      const float* dCandIn, const float* dU, const float* H, const float* U, const float* Rg,
                                      const float* Cand, const float* dH_in, int32_t dH_in_scaled, float* const* dZ, float* dW, float* db, float* dXt, float* dH,
                                      void* workspace, size_t workspace_bytes,
                                      int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, int32_t cin, void* stream) {
    if (int rc = check_dims("stc_cell_gates_bwd_f32", Ks, Kc, C, L, Lw, 2 * h, nodes)) return rc;
    STC_REQUIRE(cin >= 0 && cin + h <= L, STC_EINVAL, "stc_cell_gates_bwd_f32: cin=%d + h=%d exceed the row width L=%d", cin, h, L);
    if (!stc_cell_fused_supported(Ks, Kc, C, L, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_bwd_f32: shape not on the fused path");
    STC_REQUIRE(Z && W && dZ && dW && (Kc == 1 || Tc), STC_EINVAL, "stc_cell_gates_bwd_f32: null Z/W/dZ/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = Ks * Kc * Lw * 2 * h, Ho = 2 * h;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(dCandIn && H && U && Rg && dH, STC_EINVAL, "stc_cell_gates_bwd_f32: null pointer");      // dXt may be null (not wanted)
    STC_REQUIRE((dU != nullptr) != (Cand != nullptr), STC_EINVAL, "stc_cell_gates_bwd_f32: give either dU or Cand (dU is then formed from dH_in = dHnew)");
    STC_REQUIRE(!Cand || dH_in, STC_EINVAL, "stc_cell_gates_bwd_f32: Cand needs dH_in = gradient of the new state");
    for (int n = 0; n < Ks; ++n) STC_REQUIRE(Z[n] && dZ[n], STC_EINVAL, "stc_cell_gates_bwd_f32: Z[%d]/dZ[%d] is null", n, n);
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_cell_gates_bwd_f32: workspace null or not 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(Ks, Kc, C, L, Ho, 0), STC_EINVAL,
                "stc_cell_gates_bwd_f32: workspace of %zu B is too small", workspace_bytes);
    int n_parts = 0;
    float* partial = static_cast<float*>(workspace);
    int rc = STC_NOT_HANDLED;
    if (x3_enabled()) rc = stc_cell_gates_bwd_x3(Z, Ks, Tc, W, dCandIn, dU, H, U, Rg, dH_in, Cand, dZ, dXt, dH, partial, &n_parts, db != nullptr,
                                                 nodes, C, L, Lw, cin, dH_in_scaled != 0, s);
    if (rc == STC_NOT_HANDLED) rc = stc_cell_gates_bwd_mfma(Z, Ks, Tc, W, dCandIn, dU, H, U, Rg, dH_in, Cand, dZ, dXt, dH, partial, &n_parts, db != nullptr,
                                                            nodes, C, L, Lw, cin, dH_in_scaled != 0, s);
    if (rc == STC_NOT_HANDLED) return stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_bwd_f32: operands not usable by the fused path (alignment)");
    if (rc != STC_OK) return rc;
    const int stride = nW + Ho;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}

extern "C" int stc_cell_cand_bwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc, const float* W,
                                    const float* dHnew, const float* U, const float* Cand,
                                    float* const* dZ, float* dW, float* db,
                                    void* workspace, size_t workspace_bytes,
                                    int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, void* stream) {
    if (int rc = check_dims("stc_cell_cand_bwd_f32", Ks, Kc, C, L, Lw, h, nodes)) return rc;
    if (!stc_cell_fused_supported(Ks, Kc, C, L, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_cand_bwd_f32: shape not on the fused path");
    STC_REQUIRE(Z && W && dZ && dW && (Kc == 1 || Tc), STC_EINVAL, "stc_cell_cand_bwd_f32: null Z/W/dZ/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = Ks * Kc * Lw * h, Ho = h;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(dHnew && U && Cand, STC_EINVAL, "stc_cell_cand_bwd_f32: null pointer");
    for (int n = 0; n < Ks; ++n) STC_REQUIRE(Z[n] && dZ[n], STC_EINVAL, "stc_cell_cand_bwd_f32: Z[%d]/dZ[%d] is null", n, n);
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_cell_cand_bwd_f32: workspace null or not 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(Ks, Kc, C, L, Ho, 0), STC_EINVAL,
                "stc_cell_cand_bwd_f32: workspace of %zu B is too small", workspace_bytes);
    int n_parts = 0;
    float* partial = static_cast<float*>(workspace);
    int rc = STC_NOT_HANDLED;
    if (x3_enabled()) rc = stc_cell_cand_bwd_x3(Z, Ks, Tc, W, dHnew, U, Cand, dZ, partial, &n_parts, db != nullptr, nodes, C, L, Lw, s);
    if (rc == STC_NOT_HANDLED) rc = stc_cell_cand_bwd_mfma(Z, Ks, Tc, W, dHnew, U, Cand, dZ, partial, &n_parts, db != nullptr, nodes, C, L, Lw, s);
    if (rc == STC_NOT_HANDLED) return stc::fail(STC_EUNSUPPORTED, "stc_cell_cand_bwd_f32: operands not usable by the fused path (alignment)");
    if (rc != STC_OK) return rc;
    const int stride = nW + Ho;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}

// ---- planar cell inputs: [Xt | H] as two contiguous (nodes, C, h) planes (Ks = Kc = 2, cin = h = 16)
extern "C" int stc_cell_planar_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t h) {
    return (x3_enabled() && Ks == 2 && Kc == 2 && (C == 32 || C == 64) && h == 16) ? 1 : 0;
}

extern "C" int stc_cell_gates_fwd_planar_f32(const float* X, const float* H, const float* SX, const float* SH,
                                             const float* Tc, const float* W, const float* bias,
                                             float* U, float* Rg, float* RH,
                                             const float* Wc, const float* bc, float* A, float* Bm,
                                             int32_t operand_format, float* act_amax,
                                             int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    if (int rc = check_dims("stc_cell_gates_fwd_planar_f32", 2, 2, C, Lw == 2 * h ? 2 * h : 20, Lw, 2 * h, nodes)) return rc;
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL,
                "stc_cell_gates_fwd_planar_f32: operand_format %d (STC_FMT_BF16X3 or STC_FMT_F16X2)", operand_format);
    STC_REQUIRE(Lw == 2 * h || (Lw > h && Lw <= h + 4), STC_EINVAL, "stc_cell_gates_fwd_planar_f32: input width %d (Lw - h) must be h or 1..4", Lw - h);
    if (!stc_cell_planar_supported(2, 2, C, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_fwd_planar_f32: shape not on the planar path");
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(X && H && SX && SH && Tc && W && U && Rg && (RH || A), STC_EINVAL, "stc_cell_gates_fwd_planar_f32: null pointer");
    STC_REQUIRE((A == nullptr) == (Bm == nullptr) && (!A || Wc), STC_EINVAL, "stc_cell_gates_fwd_planar_f32: A, Bm and Wc go together");
    const int rc = stc_cell_gates_fwd_planar_x3(X, H, SX, SH, Tc, W, bias, U, Rg, RH, Wc, bc, A, Bm, operand_format, act_amax, nodes, C, Lw, static_cast<hipStream_t>(stream));
    return rc == STC_NOT_HANDLED ? stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_fwd_planar_f32: operands not usable (alignment)") : rc;
}

extern "C" int stc_cell_gates_bwd_planar_f32(const float* X, const float* H, const float* SX, const float* SH,
                                             const float* Tc, const float* W,
                                             const float* dCandIn, const float* Cand, const float* U, const float* Rg, const float* dHnew,
                                             float* const* dZ, float* dW, float* db, float* dH,
                                             int32_t operand_format, const float* act_amax,
                                             void* workspace, size_t workspace_bytes,
                                             int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    const int L = Lw == 2 * h ? 2 * h : 20, Ho = 2 * h;
    if (int rc = check_dims("stc_cell_gates_bwd_planar_f32", 2, 2, C, L, Lw, Ho, nodes)) return rc;
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL, "stc_cell_gates_bwd_planar_f32: operand_format %d", operand_format);
    STC_REQUIRE(Lw == 2 * h || (Lw > h && Lw <= h + 4), STC_EINVAL, "stc_cell_gates_bwd_planar_f32: input width %d (Lw - h) must be h or 1..4", Lw - h);
    if (!stc_cell_planar_supported(2, 2, C, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_bwd_planar_f32: shape not on the planar path");
    STC_REQUIRE(W && dZ && dW && Tc, STC_EINVAL, "stc_cell_gates_bwd_planar_f32: null W/dZ/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = 4 * Lw * Ho;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(X && H && SX && SH && dCandIn && Cand && U && Rg && dHnew && dZ[2] && dZ[3] && (Lw != 2 * h || (dZ[0] && dZ[1])), STC_EINVAL,
                "stc_cell_gates_bwd_planar_f32: null pointer");       // (dH may be null: its values are then folded into dZ[2])
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_cell_gates_bwd_planar_f32: workspace null or not 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(2, 2, C, L, Ho, 0), STC_EINVAL,
                "stc_cell_gates_bwd_planar_f32: workspace of %zu B is too small", workspace_bytes);
    int n_parts = 0;
    float* partial = static_cast<float*>(workspace);
    const int rc = stc_cell_gates_bwd_planar_x3(X, H, SX, SH, Tc, W, dCandIn, Cand, U, Rg, dHnew, dZ, dH, partial, &n_parts, db != nullptr,
                                                operand_format, act_amax, nodes, C, Lw, s);
    if (rc == STC_NOT_HANDLED) return stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_bwd_planar_f32: operands not usable (alignment)");
    if (rc != STC_OK) return rc;
    const int stride = nW + Ho;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}

// ---- the whole backward of a planar cell step in one launch (stc_cell_bwd_x3.hip)
extern "C" int stc_cell_bwd_planar_supported(int32_t C, int32_t h) {
    return (x3_enabled() && stc_cell_bwd_planar_shape_ok(C, h)) ? 1 : 0;
}

extern "C" size_t stc_cell_bwd_planar_workspace_bytes(int32_t C, int32_t Lw, int32_t h) {
    const int L = Lw == 2 * h ? 2 * h : 20;
    return stc_bdg_node_bwd_workspace_bytes(2, 2, C, L, 2 * h, 0) + stc_bdg_node_bwd_workspace_bytes(2, 2, C, L, h, 0);
}

extern "C" int stc_cell_bwd_planar_f32(const float* X, const float* H, const float* SX, const float* SH,
                                       const float* Tc, const float* Wg, const float* Wc,
                                       const float* U, const float* Rg, const float* Cand, const float* dHnew, const float* dBm,
                                       float* dX, float* dSX, float* dH, float* dSH,
                                       float* dWg, float* dbg, float* dWc, float* dbc,
                                       int32_t accumulate_x, int32_t accumulate_h,
                                       int32_t operand_format, const float* act_amax,
                                       void* workspace, size_t workspace_bytes,
                                       int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    const int L = Lw == 2 * h ? 2 * h : 20;
    if (int rc = check_dims("stc_cell_bwd_planar_f32", 2, 2, C, L, Lw, 2 * h, nodes)) return rc;
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL, "stc_cell_bwd_planar_f32: operand_format %d", operand_format);
    STC_REQUIRE(Lw == 2 * h || (Lw > h && Lw <= h + 4), STC_EINVAL, "stc_cell_bwd_planar_f32: input width %d (Lw - h) must be h or 1..4", Lw - h);
    if (!stc_cell_bwd_planar_supported(C, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_bwd_planar_f32: C=%d h=%d is not built (C = 32, h = 16)", C, h);
    STC_REQUIRE(Wg && Wc && dWg && dWc && Tc, STC_EINVAL, "stc_cell_bwd_planar_f32: null W/dW/Tc");
    STC_REQUIRE(!accumulate_x || Lw == 2 * h, STC_EINVAL, "stc_cell_bwd_planar_f32: accumulate_x with a narrow input plane (it gets no gradient)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nWg = 4 * Lw * 2 * h, nWc = 4 * Lw * h;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dWg, 0, (size_t)nWg * sizeof(float), s), "memset dWg")) return rc;
        if (int rc = stc::hip_status(hipMemsetAsync(dWc, 0, (size_t)nWc * sizeof(float), s), "memset dWc")) return rc;
        if (dbg) if (int rc = stc::hip_status(hipMemsetAsync(dbg, 0, (size_t)2 * h * sizeof(float), s), "memset dbg")) return rc;
        if (dbc) if (int rc = stc::hip_status(hipMemsetAsync(dbc, 0, (size_t)h * sizeof(float), s), "memset dbc")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(X && H && SX && SH && U && Rg && Cand && dHnew && dBm && dH && dSH && (Lw != 2 * h || (dX && dSX)), STC_EINVAL,
                "stc_cell_bwd_planar_f32: null pointer");
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_cell_bwd_planar_f32: workspace null or not 16-byte aligned");
    const size_t bytes_g = stc_bdg_node_bwd_workspace_bytes(2, 2, C, L, 2 * h, 0);
    STC_REQUIRE(workspace_bytes >= stc_cell_bwd_planar_workspace_bytes(C, Lw, h), STC_EINVAL,
                "stc_cell_bwd_planar_f32: workspace of %zu B is too small", workspace_bytes);
    float* partial_g = static_cast<float*>(workspace);
    float* partial_c = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + bytes_g);
    int n_parts = 0;
    const int rc = stc_cell_bwd_planar_x3(X, H, SX, SH, Tc, Wg, Wc, U, Rg, Cand, dHnew, dBm, dX, dSX, dH, dSH, partial_g, partial_c, &n_parts,
                                          dbg != nullptr, dbc != nullptr, accumulate_x != 0, accumulate_h != 0, operand_format, act_amax, nodes, C, Lw, s);
    if (rc == STC_NOT_HANDLED) return stc::fail(STC_EUNSUPPORTED, "stc_cell_bwd_planar_f32: operands not usable (alignment)");
    if (rc != STC_OK) return rc;
    if (int r2 = stc_node_reduce_partials(partial_g, n_parts, nWg, 2 * h, dWg, dbg, s)) return r2;
    return stc_node_reduce_partials(partial_c, n_parts, nWc, h, dWc, dbc, s);
}

// ---- planar cell convolutions of Chebyshev order K (= 3; K = 2 has the entry points above): see stc_cell_conv_*_planar_k_x3
extern "C" int stc_cell_planar_k_supported(int32_t K, int32_t C, int32_t h) {
    return (x3_enabled() && stc_cell_planar_k_shape_ok(K, C, h)) ? 1 : 0;
}

static int planar_k_common(const char* who, const float* const* Zx, const float* const* Zh, int K, int C, int Lw, int h, int Ho, long long nodes) {
    if (int rc = check_dims(who, K, K, C, Lw == 2 * h ? 2 * h : 20, Lw, Ho, nodes)) return rc;
    STC_REQUIRE(Lw == 2 * h || (Lw > h && Lw <= h + 4), STC_EINVAL, "%s: input width %d (Lw - h) must be h or 1..4", who, Lw - h);
    if (!stc_cell_planar_k_supported(K, C, h)) return stc::fail(STC_EUNSUPPORTED, "%s: K=%d C=%d h=%d is not on the order-K planar path", who, K, C, h);
    STC_REQUIRE(Zx && Zh, STC_EINVAL, "%s: null plane arrays", who);
    if (nodes > 0)
        for (int n = 0; n < K; ++n) STC_REQUIRE(Zx[n] && Zh[n], STC_EINVAL, "%s: plane %d is null", who, n);
    return STC_OK;
}

extern "C" int stc_cell_gates_fwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W, const float* bias,
                                               float* U, float* Rg, float* RH, int32_t operand_format, float* act_amax,
                                               int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    if (int rc = planar_k_common("stc_cell_gates_fwd_planar_k_f32", Zx, Zh, K, C, Lw, h, 2 * h, nodes)) return rc;
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL, "stc_cell_gates_fwd_planar_k_f32: operand_format %d", operand_format);
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(Tc && W && U && Rg && RH, STC_EINVAL, "stc_cell_gates_fwd_planar_k_f32: null pointer");
    const int rc = stc_cell_conv_fwd_planar_k_x3(Zx, Zh, K, Tc, W, bias, 1, Zh[0], nullptr, U, Rg, RH, nullptr, nullptr, operand_format, act_amax, nodes, C, Lw, static_cast<hipStream_t>(stream));
    return rc == STC_NOT_HANDLED ? stc::fail(STC_EUNSUPPORTED, "stc_cell_gates_fwd_planar_k_f32: operands not usable (alignment)") : rc;
}

extern "C" int stc_cell_cand_fwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W, const float* bias,
                                              const float* U, const float* H, float* Cand, float* Hnew, int32_t operand_format, float* act_amax,
                                              int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    if (int rc = planar_k_common("stc_cell_cand_fwd_planar_k_f32", Zx, Zh, K, C, Lw, h, h, nodes)) return rc;
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL, "stc_cell_cand_fwd_planar_k_f32: operand_format %d", operand_format);
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(Tc && W && U && H && Cand && Hnew, STC_EINVAL, "stc_cell_cand_fwd_planar_k_f32: null pointer");
    STC_REQUIRE(stc::aligned16(U) && stc::aligned16(H) && stc::aligned16(Cand) && stc::aligned16(Hnew), STC_EALIGN, "stc_cell_cand_fwd_planar_k_f32: misaligned operand");
    const int rc = stc_cell_conv_fwd_planar_k_x3(Zx, Zh, K, Tc, W, bias, 2, H, U, nullptr, nullptr, nullptr, Cand, Hnew, operand_format, act_amax, nodes, C, Lw, static_cast<hipStream_t>(stream));
    return rc == STC_NOT_HANDLED ? stc::fail(STC_EUNSUPPORTED, "stc_cell_cand_fwd_planar_k_f32: operands not usable (alignment)") : rc;
}

static int planar_k_bwd(const char* who, const float* const* Zx, const float* const* Zh, int K, const float* Tc, const float* W, int mode,
                        const float* dRH, const float* Cand, const float* U, const float* Rg, const float* dHnew,
                        float* const* dZx, float* const* dZh, float* dW, float* db, float* dH, void* workspace, size_t workspace_bytes,
                        long long nodes, int C, int Lw, int h, hipStream_t s, int operand_format, const float* act_amax, int accumulate_x = 0) {
    const int L = Lw == 2 * h ? 2 * h : 20, Ho = mode == 1 ? 2 * h : h;
    if (int rc = planar_k_common(who, Zx, Zh, K, C, Lw, h, Ho, nodes)) return rc;
    STC_REQUIRE(W && dW && Tc && dZh && (Lw != 2 * h || dZx), STC_EINVAL, "%s: null W/dW/Tc/dZ", who);
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL, "%s: operand_format %d", who, operand_format);
    const int nW = K * K * Lw * Ho;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(Cand && U && dHnew && (mode != 1 || (dRH && Rg)), STC_EINVAL, "%s: null pointer", who);      // (dH may be null: folded into dZh[0])
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "%s: workspace null or not 16-byte aligned", who);
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(K, K, C, L, Ho, 0), STC_EINVAL, "%s: workspace of %zu B is too small", who, workspace_bytes);
    int n_parts = 0;
    float* partial = static_cast<float*>(workspace);
    const int rc = stc_cell_conv_bwd_planar_k_x3(Zx, Zh, K, Tc, W, mode, dRH, Cand, U, Rg, dHnew, dZx, dZh, dH, partial, &n_parts, db != nullptr, nodes, C, Lw,
                                                 accumulate_x, operand_format, act_amax, s);
    if (rc == STC_NOT_HANDLED) return stc::fail(STC_EUNSUPPORTED, "%s: operands not usable (alignment / null gradient plane / accumulate_x outside the wide folded form)", who);
    if (rc != STC_OK) return rc;
    const int stride = nW + Ho;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}

extern "C" int stc_cell_gates_bwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W,
                                               const float* dRH, const float* Cand, const float* U, const float* Rg, const float* dHnew,
                                               float* const* dZx, float* const* dZh, float* dW, float* db, float* dH, int32_t accumulate_x,
                                               int32_t operand_format, const float* act_amax,
                                               void* workspace, size_t workspace_bytes, int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    return planar_k_bwd("stc_cell_gates_bwd_planar_k_f32", Zx, Zh, K, Tc, W, 1, dRH, Cand, U, Rg, dHnew, dZx, dZh, dW, db, dH, workspace, workspace_bytes,
                        nodes, C, Lw, h, static_cast<hipStream_t>(stream), operand_format, act_amax, accumulate_x);
}

extern "C" int stc_cell_cand_bwd_planar_k_f32(const float* const* Zx, const float* const* Zh, int32_t K, const float* Tc, const float* W,
                                              const float* dHnew, const float* U, const float* Cand,
                                              float* const* dZx, float* const* dZh, float* dW, float* db,
                                              int32_t operand_format, const float* act_amax,
                                              void* workspace, size_t workspace_bytes, int64_t nodes, int32_t C, int32_t Lw, int32_t h, void* stream) {
    return planar_k_bwd("stc_cell_cand_bwd_planar_k_f32", Zx, Zh, K, Tc, W, 2, nullptr, Cand, U, nullptr, dHnew, dZx, dZh, dW, db, nullptr, workspace, workspace_bytes,
                        nodes, C, Lw, h, static_cast<hipStream_t>(stream), operand_format, act_amax);
}

extern "C" int stc_bdg_node_post_supported(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho) {
    return (x3_enabled() && Ks == Kc && stc_node_post_shape_ok(Ks, C, L, Ho)) ? 1 : 0;
}

extern "C" int stc_bdg_node_post_fwd_f32(const float* X, const float* X2, const float* Tc, const float* W, const float* bias,
                                         float* A, float* Bm,
                                         int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    if (int rc = check_dims("stc_bdg_node_post_fwd_f32", 2, 2, C, L, Lw, Ho, nodes)) return rc;
    if (!stc_bdg_node_post_supported(2, 2, C, L, Ho)) return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_post_fwd_f32: shape not on the post-aggregation path");
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(X && Tc && W && A && Bm, STC_EINVAL, "stc_bdg_node_post_fwd_f32: null pointer");
    STC_REQUIRE(A != Bm && X != A && X != Bm, STC_EINVAL, "stc_bdg_node_post_fwd_f32: outputs must not alias");
    STC_REQUIRE(!X2 || L == 32 || (L == 20 && Lw > 16), STC_EINVAL, "stc_bdg_node_post_fwd_f32: planar input (X2) needs rows of 16 + 16 or 16 + cin (<= 4) columns, L = %d", L);
    const int rc = stc_node_post_fwd_x3(X, X2, Tc, W, bias, A, Bm, nodes, C, L, Lw, Ho, static_cast<hipStream_t>(stream));
    return rc == STC_NOT_HANDLED ? stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_post_fwd_f32: operands not usable (alignment)") : rc;
}

extern "C" int stc_bdg_node_post_bwd_f32(const float* X, const float* X2, const float* Tc, const float* W, const float* dA, const float* dB,
                                         float* dX, float* dX2, float* dW, float* db, int32_t operand_format,
                                         const float* act_amax_x, const float* act_amax_x2,
                                         void* workspace, size_t workspace_bytes,
                                         int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    if (int rc = check_dims("stc_bdg_node_post_bwd_f32", 2, 2, C, L, Lw, Ho, nodes)) return rc;
    STC_REQUIRE(operand_format == STC_FMT_BF16X3 || operand_format == STC_FMT_F16X2, STC_EINVAL, "stc_bdg_node_post_bwd_f32: operand_format %d", operand_format);
    if (!stc_bdg_node_post_supported(2, 2, C, L, Ho)) return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_post_bwd_f32: shape not on the post-aggregation path");
    STC_REQUIRE(W && dW && Tc, STC_EINVAL, "stc_bdg_node_post_bwd_f32: null W/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = 4 * Lw * Ho;
    if (nodes == 0) {
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(X && dA && dB && dX, STC_EINVAL, "stc_bdg_node_post_bwd_f32: null pointer");
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_bdg_node_post_bwd_f32: workspace null or not 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(2, 2, C, L, Ho, 0), STC_EINVAL,
                "stc_bdg_node_post_bwd_f32: workspace of %zu B is too small", workspace_bytes);
    int n_parts = 0;
    float* partial = static_cast<float*>(workspace);
    STC_REQUIRE(!X2 || L == 32 || (L == 20 && Lw > 16), STC_EINVAL, "stc_bdg_node_post_bwd_f32: planar input (X2) needs rows of 16 + 16 or 16 + cin (<= 4) columns, L = %d", L);
    STC_REQUIRE(L == 20 ? dX2 == nullptr : (X2 == nullptr) == (dX2 == nullptr), STC_EINVAL,
                "stc_bdg_node_post_bwd_f32: planar input (X2) and planar gradient (dX2) go together (the narrow input plane of L = 20 gets no gradient)");
    const int rc = stc_node_post_bwd_x3(X, X2, Tc, W, dA, dB, dX, dX2, partial, &n_parts, db != nullptr, operand_format, act_amax_x, act_amax_x2, nodes, C, L, Lw, Ho, s);
    if (rc == STC_NOT_HANDLED) return stc::fail(STC_EUNSUPPORTED, "stc_bdg_node_post_bwd_f32: operands not usable (alignment)");
    if (rc != STC_OK) return rc;
    const int stride = nW + Ho;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}

extern "C" int stc_cell_blend_fwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                                      const float* W, const float* bias, const float* U, const float* H,
                                      float* Cand, float* Hnew,
                                      float* copy0, int32_t copy0_ld, int32_t copy0_off, const float* side_src, int32_t side_cin,
                                      float* copy1, int32_t copy1_ld, int32_t copy1_off,
                                      int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t h, void* stream) {
    if (int rc = check_dims("stc_cell_blend_fwd_f32", Ks, Kc, C, L, Lw, h, nodes)) return rc;
    if (!stc_cell_fused_supported(Ks, Kc, C, L, h)) return stc::fail(STC_EUNSUPPORTED, "stc_cell_blend_fwd_f32: shape not on the fused path");
    if (nodes == 0) return STC_OK;
    STC_REQUIRE(Z && W && U && H && Cand && Hnew && (Kc == 1 || Tc), STC_EINVAL, "stc_cell_blend_fwd_f32: null pointer");
    for (int n = 0; n < Ks; ++n) STC_REQUIRE(Z[n], STC_EINVAL, "stc_cell_blend_fwd_f32: Z[%d] is null", n);
    STC_REQUIRE((!copy0 || (copy0_off >= 0 && copy0_off + h <= copy0_ld)) && (!copy1 || (copy1_off >= 0 && copy1_off + h <= copy1_ld)),
                STC_EINVAL, "stc_cell_blend_fwd_f32: state copy columns [off, off+%d) do not fit the row width", h);
    STC_REQUIRE(!side_src || (copy0 && side_cin >= 0 && side_cin == copy0_off), STC_EINVAL,
                "stc_cell_blend_fwd_f32: side_src needs copy0 with copy0_off == side_cin (the state columns follow the input columns)");
    StcStateCopies cp{};
    cp.dst[0] = copy0; cp.ld[0] = copy0_ld; cp.off[0] = copy0_off; cp.side_src = side_src; cp.side_cin = side_cin;
    cp.dst[1] = copy1; cp.ld[1] = copy1_ld; cp.off[1] = copy1_off;
    int rc = STC_NOT_HANDLED;
    if (x3_enabled()) rc = stc_cell_blend_fwd_x3(Z, Ks, Tc, W, bias, U, H, Cand, Hnew, &cp, nodes, C, L, Lw, static_cast<hipStream_t>(stream));
    if (rc == STC_NOT_HANDLED) rc = stc_cell_blend_fwd_mfma(Z, Ks, Tc, W, bias, U, H, Cand, Hnew, &cp, nodes, C, L, Lw, static_cast<hipStream_t>(stream));
    return rc == STC_NOT_HANDLED ? stc::fail(STC_EUNSUPPORTED, "stc_cell_blend_fwd_f32: operands not usable by the fused path (alignment)") : rc;
}

extern "C" int stc_set_dispatch_level(int32_t level) {
    STC_REQUIRE(level >= 0 && level <= 2, STC_EINVAL, "stc_set_dispatch_level: level %d (0 = all paths, 1 = no split-operand kernels, 2 = generic kernels only)", level);
    g_dispatch_level.store(level, std::memory_order_relaxed);
    return STC_OK;
}

extern "C" size_t stc_bdg_node_bwd_workspace_bytes(int32_t Ks, int32_t Kc, int32_t C, int32_t L, int32_t Ho,
                                                   int32_t /*want_dTc*/) {
    if (Ks < 1 || Kc < 1 || C < 1 || L < 1 || Ho < 1) return 0;
    const size_t per = (size_t)Ks * Kc * L * Ho + Ho + (size_t)Kc * C * C;
    return per * sizeof(float) * NODE_BWD_MAX_GRID;
}

extern "C" int stc_bdg_node_bwd_f32(const float* const* Z, int32_t Ks, const float* Tc, int32_t Kc,
                                    const float* W, const float* dY,
                                    float* const* dZ, float* dW, float* db, float* dTc,
                                    void* workspace, size_t workspace_bytes,
                                    int64_t nodes, int32_t C, int32_t L, int32_t Lw, int32_t Ho, void* stream) {
    if (int rc = check_dims("stc_bdg_node_bwd_f32", Ks, Kc, C, L, Lw, Ho, nodes)) return rc;
    STC_REQUIRE(Z && W && dZ && dW && (Kc == 1 || Tc), STC_EINVAL, "stc_bdg_node_bwd_f32: null Z/W/dZ/dW/Tc");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nW = Ks * Kc * Lw * Ho, nT = Kc * C * C;
    if (nodes == 0) {   // gradients of an empty batch are zero
        if (int rc = stc::hip_status(hipMemsetAsync(dW, 0, (size_t)nW * sizeof(float), s), "memset dW")) return rc;
        if (db) if (int rc = stc::hip_status(hipMemsetAsync(db, 0, (size_t)Ho * sizeof(float), s), "memset db")) return rc;
        if (dTc) if (int rc = stc::hip_status(hipMemsetAsync(dTc, 0, (size_t)nT * sizeof(float), s), "memset dTc")) return rc;
        return STC_OK;
    }
    STC_REQUIRE(dY, STC_EINVAL, "stc_bdg_node_bwd_f32: null dY");
    ZPtrs zp{};
    DZPtrs dzp{};
    for (int n = 0; n < Ks; ++n) {
        STC_REQUIRE(Z[n] && dZ[n], STC_EINVAL, "stc_bdg_node_bwd_f32: Z[%d]/dZ[%d] is null", n, n);
        zp.p[n] = Z[n];
        dzp.p[n] = dZ[n];
    }
    STC_REQUIRE(workspace && stc::aligned16(workspace), STC_EALIGN, "stc_bdg_node_bwd_f32: workspace null or not 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_bdg_node_bwd_workspace_bytes(Ks, Kc, C, L, Ho, dTc != nullptr), STC_EINVAL,
                "stc_bdg_node_bwd_f32: workspace of %zu B is too small", workspace_bytes);
    if (mfma_enabled() && dTc == nullptr) {
        int n_parts = 0;
        float* partial = static_cast<float*>(workspace);
        int rc = STC_NOT_HANDLED;
        if (x3_enabled()) rc = stc_node_bwd_x3(Z, Ks, Tc, Kc, W, dY, dZ, partial, &n_parts, db != nullptr, nodes, C, L, Lw, Ho, s);
        if (rc == STC_NOT_HANDLED) rc = stc_node_bwd_mfma(Z, Ks, Tc, Kc, W, dY, dZ, partial, &n_parts, db != nullptr, nodes, C, L, Lw, Ho, s);
        if (rc == STC_OK) {
            const int stride = nW + Ho;
            hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                               partial, n_parts, stride, nW, Ho, 0, dW, db, static_cast<float*>(nullptr));
            STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
            return STC_OK;
        }
        if (rc != STC_NOT_HANDLED) return rc;
    }
    const NodeDims d = make_dims(Ks, Kc, C, L, Lw, Ho);
    const bool want_dT = dTc != nullptr && Kc > 1;
    const BwdCarve cv = bwd_carve(d, want_dT);
    const size_t lds = (size_t)cv.total * sizeof(float);
    STC_REQUIRE(lds <= stc::kMaxLdsBytes, STC_ELIMIT,
                "stc_bdg_node_bwd_f32: needs %zu B of LDS (C=%d L=%d Ho=%d Ks=%d Kc=%d), limit %zu", lds, C, L, Ho, Ks, Kc, stc::kMaxLdsBytes);
    if (int rc = stc::hip_status(stc::allow_lds(bdg_node_bwd_kernel, lds), "hipFuncSetAttribute(node bwd)")) return rc;
    const long long n_tiles = (nodes + d.TN - 1) / d.TN;
    const int grid = bwd_grid(d, nodes);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(bdg_node_bwd_kernel, dim3(grid), dim3(NODE_THREADS), lds, s,
                       zp, Tc, W, dY, dzp, partial, (long long)nodes * C, d, cv, (int)n_tiles, (int)want_dT, (int)(db != nullptr));
    STC_LAUNCH_CHECK("stc_bdg_node_bwd_f32 launch");
    const int stride = nW + Ho + nT;
    hipLaunchKernelGGL(bdg_node_reduce_kernel, dim3((stride + RED_ELEMS - 1) / RED_ELEMS), dim3(NODE_THREADS), 0, s,
                       partial, grid, stride, nW, Ho, nT, dW, db, dTc);
    STC_LAUNCH_CHECK("stc_bdg_node_reduce launch");
    return STC_OK;
}
