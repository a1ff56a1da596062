// GRU gate math of STC_Cell (STC_GNN.py:68-78), the matrix-side Chebyshev set of the
// small category graph (STC_GNN.py:24-29) and the pointwise helpers -- gfx950.
// All of these are streaming, HBM-bound kernels: one pass over each operand.
#include "stc_common.h"

#include <initializer_list>

namespace {

constexpr int EW_THREADS = 256;

inline dim3 ew_grid(long long n) {
    long long blocks = (n + EW_THREADS - 1) / EW_THREADS;
    const long long cap = (long long)stc::kNumCu * 8;     // grid-stride beyond 2048 blocks
    return dim3((unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---- gates: U = sigmoid(G[:, :h]); Rg = sigmoid(G[:, h:]); CandIn = [Xt | Rg*H]
__global__ __launch_bounds__(EW_THREADS) void gru_gates_fwd_kernel(
    const float* __restrict__ G, const float* __restrict__ Xt, const float* __restrict__ H,
    float* __restrict__ U, float* __restrict__ Rg, float* __restrict__ CandIn,
    long long rows, int cin, int h, int pad) {
    const int L = cin + h + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        if (l < cin) {
            CandIn[e] = Xt[r * cin + l];
        } else if (l >= cin + h) {
            CandIn[e] = 0.f;
        } else {
            const int k = l - cin;
            const float u = sigmoidf_(G[r * 2 * h + k]);
            const float g = sigmoidf_(G[r * 2 * h + h + k]);
            U[r * h + k] = u;
            Rg[r * h + k] = g;
            CandIn[e] = g * H[r * h + k];
        }
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_gates_bwd_kernel(
    const float* __restrict__ dCandIn, const float* __restrict__ dU, const float* __restrict__ H,
    const float* __restrict__ U, const float* __restrict__ Rg, const float* dH_in,
    float* __restrict__ dG, float* __restrict__ dXt, float* dH,
    long long rows, int cin, int h, int pad) {
    const int L = cin + h + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        if (l < cin) {
            dXt[r * cin + l] = dCandIn[e];
        } else if (l < cin + h) {
            const int k = l - cin;
            const long long i = r * h + k;
            const float u = U[i], g = Rg[i], d = dCandIn[e];
            dG[r * 2 * h + k] = dU[i] * u * (1.f - u);
            dG[r * 2 * h + h + k] = d * H[i] * g * (1.f - g);
            dH[i] = dH_in ? fmaf(d, g, dH_in[i]) : d * g;
        }
    }
}

// ---- blend: Cand = tanh(Cpre); Hnew = (1-U)*H + U*Cand
__global__ __launch_bounds__(EW_THREADS) void gru_blend_fwd_kernel(
    const float* __restrict__ Cpre, const float* __restrict__ U, const float* __restrict__ H,
    float* __restrict__ Cand, float* __restrict__ Hnew, long long n) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const float c = tanhf(Cpre[e]);
        const float u = U[e];
        Cand[e] = c;
        Hnew[e] = (1.f - u) * H[e] + u * c;
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_blend_bwd_kernel(
    const float* __restrict__ dHnew, const float* __restrict__ U, const float* __restrict__ H,
    const float* __restrict__ Cand, float* __restrict__ dCpre, float* __restrict__ dU, float* __restrict__ dH,
    long long n) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const float g = dHnew[e], u = U[e], c = Cand[e];
        dCpre[e] = g * u * (1.f - c * c);
        if (dU) dU[e] = g * (c - H[e]);
        if (dH) dH[e] = g * (1.f - u);
    }
}


// ---- 16-byte versions (every width a multiple of 4 floats, pointers 16-byte aligned): one float4 per thread
__device__ __forceinline__ float4 sig4(float4 x) { return make_float4(sigmoidf_(x.x), sigmoidf_(x.y), sigmoidf_(x.z), sigmoidf_(x.w)); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

__global__ __launch_bounds__(EW_THREADS) void gru_gates_fwd_vec_kernel(
    const float4* __restrict__ G, const float4* __restrict__ Xt, const float4* __restrict__ H,
    float4* __restrict__ U, float4* __restrict__ Rg, float4* __restrict__ CandIn,
    long long rows, int cin4, int h4, int pad4) {
    const int L4 = cin4 + h4 + pad4;
    const long long n = rows * L4;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L4;
        const int l = (int)(e - r * L4);
        if (l < cin4) {
            CandIn[e] = Xt[r * cin4 + l];
        } else if (l >= cin4 + h4) {
            CandIn[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const int k = l - cin4;
            const float4 u = sig4(G[r * 2 * h4 + k]);
            const float4 g = sig4(G[r * 2 * h4 + h4 + k]);
            U[r * h4 + k] = u;
            Rg[r * h4 + k] = g;
            CandIn[e] = mul4(g, H[r * h4 + k]);
        }
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_gates_bwd_vec_kernel(
    const float4* __restrict__ dCandIn, const float4* __restrict__ dU, const float4* __restrict__ H,
    const float4* __restrict__ U, const float4* __restrict__ Rg, const float4* dH_in,
    float4* __restrict__ dG, float4* __restrict__ dXt, float4* dH,
    long long rows, int cin4, int h4, int pad4) {
    const int L4 = cin4 + h4 + pad4;
    const long long n = rows * L4;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L4;
        const int l = (int)(e - r * L4);
        if (l < cin4) {
            dXt[r * cin4 + l] = dCandIn[e];
        } else if (l < cin4 + h4) {
            const int k = l - cin4;
            const long long i = r * h4 + k;
            const float4 u = U[i], g = Rg[i], d = dCandIn[e], du = dU[i], hh = H[i];
            dG[r * 2 * h4 + k] = make_float4(du.x * u.x * (1.f - u.x), du.y * u.y * (1.f - u.y), du.z * u.z * (1.f - u.z), du.w * u.w * (1.f - u.w));
            dG[r * 2 * h4 + h4 + k] = make_float4(d.x * hh.x * g.x * (1.f - g.x), d.y * hh.y * g.y * (1.f - g.y),
                                                  d.z * hh.z * g.z * (1.f - g.z), d.w * hh.w * g.w * (1.f - g.w));
            float4 o = mul4(d, g);
            if (dH_in) { const float4 p = dH_in[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
            dH[i] = o;
        }
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_blend_fwd_vec_kernel(
    const float4* __restrict__ Cpre, const float4* __restrict__ U, const float4* __restrict__ H,
    float4* __restrict__ Cand, float4* __restrict__ Hnew, long long n4) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n4; e += (long long)gridDim.x * EW_THREADS) {
        const float4 p = Cpre[e], u = U[e], hh = H[e];
        const float4 c = make_float4(tanhf(p.x), tanhf(p.y), tanhf(p.z), tanhf(p.w));
        Cand[e] = c;
        Hnew[e] = make_float4((1.f - u.x) * hh.x + u.x * c.x, (1.f - u.y) * hh.y + u.y * c.y,
                              (1.f - u.z) * hh.z + u.z * c.z, (1.f - u.w) * hh.w + u.w * c.w);
    }
}

__global__ __launch_bounds__(EW_THREADS) void gru_blend_bwd_vec_kernel(
    const float4* __restrict__ dHnew, const float4* __restrict__ U, const float4* __restrict__ H,
    const float4* __restrict__ Cand, float4* __restrict__ dCpre, float4* __restrict__ dU, float4* __restrict__ dH,
    long long n4) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n4; e += (long long)gridDim.x * EW_THREADS) {
        const float4 g = dHnew[e], u = U[e], c = Cand[e];
        dCpre[e] = make_float4(g.x * u.x * (1.f - c.x * c.x), g.y * u.y * (1.f - c.y * c.y),
                               g.z * u.z * (1.f - c.z * c.z), g.w * u.w * (1.f - c.w * c.w));
        if (dU) {
            const float4 hh = H[e];
            dU[e] = make_float4(g.x * (c.x - hh.x), g.y * (c.y - hh.y), g.z * (c.z - hh.z), g.w * (c.w - hh.w));
        }
        if (dH) dH[e] = make_float4(g.x * (1.f - u.x), g.y * (1.f - u.y), g.z * (1.f - u.z), g.w * (1.f - u.w));
    }
}

__global__ __launch_bounds__(EW_THREADS) void concat2_vec_kernel(const float4* __restrict__ A, const float4* __restrict__ B,
                                                                  float4* __restrict__ out, long long rows, int a4, int b4, int pad4) {
    const int L4 = a4 + b4 + pad4;
    const long long n = rows * L4;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L4;
        const int l = (int)(e - r * L4);
        out[e] = l < a4 ? A[r * a4 + l] : (l < a4 + b4 ? B[r * b4 + (l - a4)] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(EW_THREADS) void split2_vec_kernel(const float4* __restrict__ src, const float4* addA, const float4* addB,
                                                                 float4* A, float4* B, long long rows, int a4, int b4, int pad4, int ldA4,
                                                                 const float4* addA2, const float4* addB2) {
    const int L4 = a4 + b4 + pad4;
    const long long n = rows * L4;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L4;
        const int l = (int)(e - r * L4);
        if (l < a4) {
            if (A) {
                const long long i = r * a4 + l;
                float4 v = addA ? add4(src[e], addA[r * ldA4 + l]) : src[e];
                if (addA2) v = add4(v, addA2[i]);
                A[i] = v;
            }
        } else if (l < a4 + b4) {
            if (B) {
                const long long i = r * b4 + (l - a4);
                float4 v = addB ? add4(src[e], addB[i]) : src[e];
                if (addB2) v = add4(v, addB2[i]);
                B[i] = v;
            }
        }
    }
}

inline bool vec_ok(std::initializer_list<const void*> ptrs, std::initializer_list<long long> widths) {
    for (const void* p : ptrs)
        if (p && !stc::aligned16(p)) return false;
    for (long long w : widths)
        if (w % 4) return false;
    return true;
}
#define F4C(p) reinterpret_cast<const float4*>(p)
#define F4M(p) reinterpret_cast<float4*>(p)

// ---- output head: y = sigmoid(<H[r,:], w> + b)  (the reference's two bias-ful Linears folded into one map)
constexpr int HEAD_MAX_H = 64;
constexpr int HEAD_PARTS = 1024;

// LPR lanes share a row of h floats, one 16-byte piece each: a wave instruction then moves 1 KiB of CONTIGUOUS memory
// (one thread per row made every instruction touch 64 different 64-byte rows: 2.6 TB/s instead of ~5).
template <int LPR>
__global__ __launch_bounds__(EW_THREADS) void head_fwd_kernel(const float* __restrict__ H, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y,
                                                               long long rows, int h) {
    constexpr int RPB = EW_THREADS / LPR;              // rows per workgroup pass
    const int h4 = h / 4, q = threadIdx.x % LPR;
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < h4) wq = reinterpret_cast<const float4*>(w)[q];
    const float bias = b[0];
    for (long long r = (long long)blockIdx.x * RPB + threadIdx.x / LPR; r < rows; r += (long long)gridDim.x * RPB) {
        float s = 0.f;
        if (q < h4) {
            const float4 v = reinterpret_cast<const float4*>(H + r * h)[q];
            s = fmaf(v.x, wq.x, fmaf(v.y, wq.y, fmaf(v.z, wq.z, v.w * wq.w)));
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (q == 0) y[r] = sigmoidf_(s + bias);
    }
}

// dH = g * w (streamed); per-workgroup partial sums of dw | db in a fixed order -> workspace[block][h+1]
template <int LPR>
__global__ __launch_bounds__(EW_THREADS) void head_bwd_kernel(const float* __restrict__ H, const float* __restrict__ w,
                                                               const float* __restrict__ y, const float* __restrict__ dy,
                                                               float* __restrict__ dH, float* __restrict__ partial,
                                                               long long rows, int h) {
    constexpr int RPB = EW_THREADS / LPR;
    __shared__ float red[EW_THREADS][5];
    const int h4 = h / 4, q = threadIdx.x % LPR;
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < h4) wq = reinterpret_cast<const float4*>(w)[q];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accg = 0.f;
    for (long long r = (long long)blockIdx.x * RPB + threadIdx.x / LPR; r < rows; r += (long long)gridDim.x * RPB) {
        const float yy = y[r];
        const float g = dy[r] * yy * (1.f - yy);
        if (q == 0) accg += g;
        if (q < h4) {
            const float4 v = reinterpret_cast<const float4*>(H + r * h)[q];
            reinterpret_cast<float4*>(dH + r * h)[q] = make_float4(g * wq.x, g * wq.y, g * wq.z, g * wq.w);
            acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y); acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
        }
    }
    red[threadIdx.x][0] = acc.x; red[threadIdx.x][1] = acc.y; red[threadIdx.x][2] = acc.z; red[threadIdx.x][3] = acc.w;
    red[threadIdx.x][4] = accg;
    __syncthreads();
    if ((int)threadIdx.x <= h) {                      // element k of dw (k < h) or db (k == h): its lanes in a fixed order
        const int k = threadIdx.x;
        const int lane_q = k < h ? k / 4 : 0, comp = k < h ? k % 4 : 4;
        float v = 0.f;
        for (int t = lane_q; t < EW_THREADS; t += LPR) v += red[t][comp];
        partial[(size_t)blockIdx.x * (h + 1) + k] = v;
    }
}

// bf16 state rows, hidden 16 (the bf16-storage configuration): a row is 32 bytes, two lanes share it with one 16-byte piece of
// 8 bf16 each; y, dy and the weight gradient stay fp32
using head_u4 = __attribute__((ext_vector_type(4))) unsigned;
using head_bf2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ void head_unpack8(const head_u4 v, float (&r)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[2 * i] = __uint_as_float(v[i] << 16); r[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
}
__device__ __forceinline__ unsigned head_pk(float a, float b) {
    const head_bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(EW_THREADS) void head_fwd_bf16_kernel(const head_u4* __restrict__ H, const float* __restrict__ w,
                                                                    const float* __restrict__ b, float* __restrict__ y, long long rows) {
    constexpr int RPB = EW_THREADS / 2;
    const int q = threadIdx.x & 1;
    float wq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wq[i] = w[8 * q + i];
    const float bias = b[0];
    for (long long r = (long long)blockIdx.x * RPB + threadIdx.x / 2; r < rows; r += (long long)gridDim.x * RPB) {
        float v[8];
        head_unpack8(H[2 * r + q], v);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s = fmaf(v[i], wq[i], s);
        s += __shfl_xor(s, 1, 64);
        if (q == 0) y[r] = sigmoidf_(s + bias);
    }
}

__global__ __launch_bounds__(EW_THREADS) void head_bwd_bf16_kernel(const head_u4* __restrict__ H, const float* __restrict__ w,
                                                                    const float* __restrict__ y, const float* __restrict__ dy,
                                                                    head_u4* __restrict__ dH, float* __restrict__ partial, long long rows) {
    constexpr int RPB = EW_THREADS / 2, h = 16;
    __shared__ float red[EW_THREADS][9];
    const int q = threadIdx.x & 1;
    float wq[8], acc[8], accg = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { wq[i] = w[8 * q + i]; acc[i] = 0.f; }
    for (long long r = (long long)blockIdx.x * RPB + threadIdx.x / 2; r < rows; r += (long long)gridDim.x * RPB) {
        const float yy = y[r];
        const float g = dy[r] * yy * (1.f - yy);
        if (q == 0) accg += g;
        float v[8];
        head_unpack8(H[2 * r + q], v);
        dH[2 * r + q] = head_u4{head_pk(g * wq[0], g * wq[1]), head_pk(g * wq[2], g * wq[3]), head_pk(g * wq[4], g * wq[5]), head_pk(g * wq[6], g * wq[7])};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(g, v[i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
    red[threadIdx.x][8] = accg;
    __syncthreads();
    if ((int)threadIdx.x <= h) {                      // element k of dw (k < 16) or db (k == 16): its lanes in a fixed order
        const int k = threadIdx.x;
        const int lane_q = k < h ? k / 8 : 0, comp = k < h ? k % 8 : 8;
        float v = 0.f;
        for (int t = lane_q; t < EW_THREADS; t += 2) v += red[t][comp];
        partial[(size_t)blockIdx.x * (h + 1) + k] = v;
    }
}

__global__ __launch_bounds__(EW_THREADS) void head_reduce_kernel(const float* __restrict__ partial, int n_parts, int stride, float* __restrict__ out) {
    __shared__ float red[EW_THREADS];
    const int e = blockIdx.x;                       // one output element per workgroup
    float s = 0.f;
    for (int p = threadIdx.x; p < n_parts; p += EW_THREADS) s += partial[(size_t)p * stride + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = EW_THREADS / 2; off > 0; off >>= 1) {      // fixed tree: bitwise reproducible
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[e] = red[0];
}

__global__ __launch_bounds__(EW_THREADS) void axpy_kernel(float a, const float* __restrict__ x, float* y, long long n) {
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS)
        y[e] = fmaf(a, x[e], y[e]);
}

__global__ __launch_bounds__(EW_THREADS) void concat2_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ out, long long rows, int a, int b, int pad) {
    const int L = a + b + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        out[e] = l < a ? A[r * a + l] : (l < a + b ? B[r * b + (l - a)] : 0.f);
    }
}

__global__ __launch_bounds__(EW_THREADS) void split2_kernel(const float* __restrict__ src, const float* addA, const float* addB,
                                                             float* A, float* B, long long rows, int a, int b, int pad, int ldA,
                                                             const float* addA2, const float* addB2) {
    const int L = a + b + pad;
    const long long n = rows * L;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * EW_THREADS) {
        const long long r = e / L;
        const int l = (int)(e - r * L);
        if (l < a) {
            if (A) {
                const long long i = r * a + l;
                float v = addA ? src[e] + addA[r * ldA + l] : src[e];
                if (addA2) v += addA2[i];
                A[i] = v;
            }
        } else if (l < a + b) {
            if (B) {
                const long long i = r * b + (l - a);
                float v = addB ? src[e] + addB[i] : src[e];
                if (addB2) v += addB2[i];
                B[i] = v;
            }
        }
    }
}

// ---- dense Chebyshev set of the category graph (n <= 128, a handful of n x n products)
__global__ __launch_bounds__(EW_THREADS) void cheby_init_kernel(const float* __restrict__ G, int n, int K, float* __restrict__ T) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    T[e] = i == j ? 1.f : 0.f;
    if (K > 1) T[n * n + e] = G[e];
}

// Tk = (2G) Tkm1 - Tkm2, same operand order as the reference's torch.mm(2*G, T[-1]) - T[-2]
__global__ __launch_bounds__(EW_THREADS) void cheby_step_kernel(const float* __restrict__ G, const float* __restrict__ Tkm1,
                                                                 const float* __restrict__ Tkm2, float* __restrict__ Tk, int n) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    float s = 0.f;
    for (int m = 0; m < n; ++m) s = fmaf(2.f * G[i * n + m], Tkm1[m * n + j], s);
    Tk[e] = s - Tkm2[e];
}

// one reverse step k: dG += 2 dTk Tkm1^T ; dTkm1 += 2 G^T dTk ; dTkm2 -= dTk
__global__ __launch_bounds__(EW_THREADS) void cheby_bwd_step_kernel(const float* __restrict__ G, const float* __restrict__ Tkm1,
                                                                     const float* __restrict__ dTk, float* dTkm1, float* dTkm2,
                                                                     float* dG, int n) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    float sg = 0.f, st = 0.f;
    for (int m = 0; m < n; ++m) {
        sg = fmaf(dTk[i * n + m], Tkm1[j * n + m], sg);
        st = fmaf(G[m * n + i], dTk[m * n + j], st);
    }
    dG[e] += 2.f * sg;
    dTkm1[e] += 2.f * st;
    dTkm2[e] -= dTk[e];
}

__global__ __launch_bounds__(EW_THREADS) void cheby_bwd_final_kernel(const float* __restrict__ dT1, float* dG, int nn, int add) {
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= nn) return;
    dG[e] = add ? dG[e] + dT1[e] : dT1[e];
}

}  // namespace

#define STC_EW_PROLOGUE(name, n_expr, ...)                                             \
    STC_REQUIRE((n_expr) >= 0, STC_EINVAL, name ": negative size");                    \
    if ((n_expr) == 0) return STC_OK;                                                  \
    STC_REQUIRE(__VA_ARGS__, STC_EINVAL, name ": null pointer");                       \
    hipStream_t s = static_cast<hipStream_t>(stream)

extern "C" int stc_gru_gates_fwd_f32(const float* G, const float* Xt, const float* H, float* U, float* Rg, float* CandIn,
                                     int64_t rows, int32_t cin, int32_t h, int32_t pad, void* stream) {
    STC_REQUIRE(cin >= 0 && h >= 1 && pad >= 0, STC_EINVAL, "stc_gru_gates_fwd_f32: bad widths cin=%d h=%d pad=%d", cin, h, pad);
    STC_EW_PROLOGUE("stc_gru_gates_fwd_f32", rows, G && H && U && Rg && CandIn && (cin == 0 || Xt));
    const long long n = (long long)rows * (cin + h + pad);
    if (vec_ok({G, Xt, H, U, Rg, CandIn}, {cin, h, pad}))
        hipLaunchKernelGGL(gru_gates_fwd_vec_kernel, ew_grid(n / 4), dim3(EW_THREADS), 0, s, F4C(G), F4C(Xt), F4C(H), F4M(U), F4M(Rg),
                           F4M(CandIn), (long long)rows, cin / 4, h / 4, pad / 4);
    else
        hipLaunchKernelGGL(gru_gates_fwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, G, Xt, H, U, Rg, CandIn, (long long)rows, cin, h, pad);
    STC_LAUNCH_CHECK("stc_gru_gates_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_gru_gates_bwd_f32(const float* dCandIn, const float* dU, const float* H, const float* U, const float* Rg,
                                     const float* dH_in, float* dG, float* dXt, float* dH,
                                     int64_t rows, int32_t cin, int32_t h, int32_t pad, void* stream) {
    STC_REQUIRE(cin >= 0 && h >= 1 && pad >= 0, STC_EINVAL, "stc_gru_gates_bwd_f32: bad widths cin=%d h=%d pad=%d", cin, h, pad);
    STC_EW_PROLOGUE("stc_gru_gates_bwd_f32", rows, dCandIn && dU && H && U && Rg && dG && dH && (cin == 0 || dXt));
    const long long n = (long long)rows * (cin + h + pad);
    if (vec_ok({dCandIn, dU, H, U, Rg, dH_in, dG, dXt, dH}, {cin, h, pad}))
        hipLaunchKernelGGL(gru_gates_bwd_vec_kernel, ew_grid(n / 4), dim3(EW_THREADS), 0, s, F4C(dCandIn), F4C(dU), F4C(H), F4C(U), F4C(Rg),
                           F4C(dH_in), F4M(dG), F4M(dXt), F4M(dH), (long long)rows, cin / 4, h / 4, pad / 4);
    else
        hipLaunchKernelGGL(gru_gates_bwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, dCandIn, dU, H, U, Rg, dH_in, dG, dXt, dH, (long long)rows, cin, h, pad);
    STC_LAUNCH_CHECK("stc_gru_gates_bwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_gru_blend_fwd_f32(const float* Cpre, const float* U, const float* H, float* Cand, float* Hnew,
                                     int64_t n, void* stream) {
    STC_EW_PROLOGUE("stc_gru_blend_fwd_f32", n, Cpre && U && H && Cand && Hnew);
    if (vec_ok({Cpre, U, H, Cand, Hnew}, {n}))
        hipLaunchKernelGGL(gru_blend_fwd_vec_kernel, ew_grid(n / 4), dim3(EW_THREADS), 0, s, F4C(Cpre), F4C(U), F4C(H), F4M(Cand), F4M(Hnew), (long long)n / 4);
    else
        hipLaunchKernelGGL(gru_blend_fwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, Cpre, U, H, Cand, Hnew, (long long)n);
    STC_LAUNCH_CHECK("stc_gru_blend_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_gru_blend_bwd_f32(const float* dHnew, const float* U, const float* H, const float* Cand,
                                     float* dCpre, float* dU, float* dH, int64_t n, void* stream) {
    STC_EW_PROLOGUE("stc_gru_blend_bwd_f32", n, dHnew && U && Cand && dCpre && (H || !dU));      // dU, dH may be null (not wanted)
    if (vec_ok({dHnew, U, H, Cand, dCpre, dU, dH}, {n}))
        hipLaunchKernelGGL(gru_blend_bwd_vec_kernel, ew_grid(n / 4), dim3(EW_THREADS), 0, s, F4C(dHnew), F4C(U), F4C(H), F4C(Cand), F4M(dCpre), F4M(dU),
                           F4M(dH), (long long)n / 4);
    else
        hipLaunchKernelGGL(gru_blend_bwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, dHnew, U, H, Cand, dCpre, dU, dH, (long long)n);
    STC_LAUNCH_CHECK("stc_gru_blend_bwd_f32 launch");
    return STC_OK;
}


extern "C" int stc_head_fwd_f32(const float* H, const float* w, const float* b, float* y, int64_t rows, int32_t h, void* stream) {
    STC_REQUIRE(h >= 4 && h <= HEAD_MAX_H && h % 4 == 0, STC_ELIMIT, "stc_head_fwd_f32: h=%d must be a multiple of 4 in [4,%d]", h, HEAD_MAX_H);
    STC_EW_PROLOGUE("stc_head_fwd_f32", rows, H && w && b && y);
    STC_REQUIRE(stc::aligned16(H), STC_EALIGN, "stc_head_fwd_f32: H not 16-byte aligned");
    STC_REQUIRE(stc::aligned16(w), STC_EALIGN, "stc_head_fwd_f32: w not 16-byte aligned");
    const int h4 = h / 4;
#define STC_HEAD_FWD(LPR_) hipLaunchKernelGGL(head_fwd_kernel<LPR_>, ew_grid(rows * LPR_), dim3(EW_THREADS), 0, s, H, w, b, y, (long long)rows, h)
    if (h4 <= 1) STC_HEAD_FWD(1); else if (h4 <= 2) STC_HEAD_FWD(2); else if (h4 <= 4) STC_HEAD_FWD(4); else if (h4 <= 8) STC_HEAD_FWD(8); else STC_HEAD_FWD(16);
#undef STC_HEAD_FWD
    STC_LAUNCH_CHECK("stc_head_fwd_f32 launch");
    return STC_OK;
}

extern "C" size_t stc_head_bwd_workspace_bytes(int32_t h) { return h < 1 ? 0 : (size_t)HEAD_PARTS * (h + 1) * sizeof(float); }

extern "C" int stc_head_bwd_f32(const float* H, const float* w, const float* y, const float* dy, float* dH, float* dwb,
                                void* workspace, size_t workspace_bytes, int64_t rows, int32_t h, void* stream) {
    STC_REQUIRE(h >= 4 && h <= HEAD_MAX_H && h % 4 == 0, STC_ELIMIT, "stc_head_bwd_f32: h=%d must be a multiple of 4 in [4,%d]", h, HEAD_MAX_H);
    STC_REQUIRE(rows >= 0 && dwb, STC_EINVAL, "stc_head_bwd_f32: negative rows or null dwb");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0) return stc::hip_status(hipMemsetAsync(dwb, 0, (size_t)(h + 1) * sizeof(float), s), "memset dwb");
    STC_REQUIRE(H && w && y && dy && dH, STC_EINVAL, "stc_head_bwd_f32: null pointer");
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(dH) && workspace && stc::aligned16(workspace), STC_EALIGN,
                "stc_head_bwd_f32: H/dH/workspace must be 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_head_bwd_workspace_bytes(h), STC_EINVAL, "stc_head_bwd_f32: workspace too small");
    STC_REQUIRE(stc::aligned16(w), STC_EALIGN, "stc_head_bwd_f32: w not 16-byte aligned");
    const int h4 = h / 4;
    const int lpr = h4 <= 1 ? 1 : h4 <= 2 ? 2 : h4 <= 4 ? 4 : h4 <= 8 ? 8 : 16;
    long long blocks = (rows * lpr + EW_THREADS - 1) / EW_THREADS;
    const int grid = (int)(blocks < HEAD_PARTS ? blocks : HEAD_PARTS);
    float* partial = static_cast<float*>(workspace);
#define STC_HEAD_BWD(LPR_) hipLaunchKernelGGL(head_bwd_kernel<LPR_>, dim3(grid), dim3(EW_THREADS), 0, s, H, w, y, dy, dH, partial, (long long)rows, h)
    if (lpr == 1) STC_HEAD_BWD(1); else if (lpr == 2) STC_HEAD_BWD(2); else if (lpr == 4) STC_HEAD_BWD(4); else if (lpr == 8) STC_HEAD_BWD(8); else STC_HEAD_BWD(16);
#undef STC_HEAD_BWD
    hipLaunchKernelGGL(head_reduce_kernel, dim3(h + 1), dim3(EW_THREADS), 0, s, partial, grid, h + 1, dwb);
    STC_LAUNCH_CHECK("stc_head_bwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_head_fwd_bf16(const void* H, const float* w, const float* b, float* y, int64_t rows, int32_t h, void* stream) {
    STC_REQUIRE(h == 16, STC_EUNSUPPORTED, "stc_head_fwd_bf16: hidden width %d (built for 16)", h);
    STC_EW_PROLOGUE("stc_head_fwd_bf16", rows, H && w && b && y);
    STC_REQUIRE(stc::aligned16(H), STC_EALIGN, "stc_head_fwd_bf16: H not 16-byte aligned");
    hipLaunchKernelGGL(head_fwd_bf16_kernel, ew_grid(rows * 2), dim3(EW_THREADS), 0, s, static_cast<const head_u4*>(H), w, b, y, (long long)rows);
    STC_LAUNCH_CHECK("stc_head_fwd_bf16 launch");
    return STC_OK;
}

extern "C" int stc_head_bwd_bf16(const void* H, const float* w, const float* y, const float* dy, void* dH, float* dwb,
                                 void* workspace, size_t workspace_bytes, int64_t rows, int32_t h, void* stream) {
    STC_REQUIRE(h == 16, STC_EUNSUPPORTED, "stc_head_bwd_bf16: hidden width %d (built for 16)", h);
    STC_REQUIRE(rows >= 0 && dwb, STC_EINVAL, "stc_head_bwd_bf16: negative rows or null dwb");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0) return stc::hip_status(hipMemsetAsync(dwb, 0, (size_t)(h + 1) * sizeof(float), s), "memset dwb");
    STC_REQUIRE(H && w && y && dy && dH, STC_EINVAL, "stc_head_bwd_bf16: null pointer");
    STC_REQUIRE(stc::aligned16(H) && stc::aligned16(dH) && workspace && stc::aligned16(workspace), STC_EALIGN,
                "stc_head_bwd_bf16: H/dH/workspace must be 16-byte aligned");
    STC_REQUIRE(workspace_bytes >= stc_head_bwd_workspace_bytes(h), STC_EINVAL, "stc_head_bwd_bf16: workspace too small");
    long long blocks = (rows * 2 + EW_THREADS - 1) / EW_THREADS;
    const int grid = (int)(blocks < HEAD_PARTS ? blocks : HEAD_PARTS);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(head_bwd_bf16_kernel, dim3(grid), dim3(EW_THREADS), 0, s, static_cast<const head_u4*>(H), w, y, dy,
                       static_cast<head_u4*>(dH), partial, (long long)rows);
    hipLaunchKernelGGL(head_reduce_kernel, dim3(h + 1), dim3(EW_THREADS), 0, s, partial, grid, h + 1, dwb);
    STC_LAUNCH_CHECK("stc_head_bwd_bf16 launch");
    return STC_OK;
}

extern "C" int stc_axpy_f32(float a, const float* x, float* y, int64_t n, void* stream) {
    STC_EW_PROLOGUE("stc_axpy_f32", n, x && y);
    hipLaunchKernelGGL(axpy_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, a, x, y, (long long)n);
    STC_LAUNCH_CHECK("stc_axpy_f32 launch");
    return STC_OK;
}

extern "C" int stc_concat2_f32(const float* A, const float* B, float* out, int64_t rows, int32_t a, int32_t b, int32_t pad, void* stream) {
    STC_REQUIRE(a >= 0 && b >= 0 && pad >= 0, STC_EINVAL, "stc_concat2_f32: negative width");
    const long long n = (long long)rows * (a + b + pad);
    STC_EW_PROLOGUE("stc_concat2_f32", n, out && (a == 0 || A) && (b == 0 || B));
    if (vec_ok({A, B, out}, {a, b, pad}))
        hipLaunchKernelGGL(concat2_vec_kernel, ew_grid(n / 4), dim3(EW_THREADS), 0, s, F4C(A), F4C(B), F4M(out), (long long)rows, a / 4, b / 4, pad / 4);
    else
        hipLaunchKernelGGL(concat2_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, A, B, out, (long long)rows, a, b, pad);
    STC_LAUNCH_CHECK("stc_concat2_f32 launch");
    return STC_OK;
}

extern "C" int stc_split2_f32(const float* src, const float* addA, const float* addB, float* A, float* B,
                              int64_t rows, int32_t a, int32_t b, int32_t pad, int32_t addA_ld,
                              const float* addA2, const float* addB2, void* stream) {
    STC_REQUIRE(a >= 0 && b >= 0 && pad >= 0, STC_EINVAL, "stc_split2_f32: negative width");
    STC_REQUIRE(addA_ld == 0 || addA_ld >= a, STC_EINVAL, "stc_split2_f32: addA_ld=%d is smaller than the width a=%d", addA_ld, a);
    const int ldA = addA_ld ? addA_ld : a;
    const long long n = (long long)rows * (a + b + pad);
    STC_EW_PROLOGUE("stc_split2_f32", n, src && (A || B));
    if (vec_ok({src, addA, addB, A, B, addA2, addB2}, {a, b, pad, ldA}))
        hipLaunchKernelGGL(split2_vec_kernel, ew_grid(n / 4), dim3(EW_THREADS), 0, s, F4C(src), F4C(addA), F4C(addB), F4M(A), F4M(B),
                           (long long)rows, a / 4, b / 4, pad / 4, ldA / 4, F4C(addA2), F4C(addB2));
    else
        hipLaunchKernelGGL(split2_kernel, ew_grid(n), dim3(EW_THREADS), 0, s, src, addA, addB, A, B, (long long)rows, a, b, pad, ldA, addA2, addB2);
    STC_LAUNCH_CHECK("stc_split2_f32 launch");
    return STC_OK;
}

extern "C" int stc_cheby_dense_fwd_f32(const float* G, int32_t n, int32_t K, float* T, void* stream) {
    STC_REQUIRE(n >= 1 && n <= 128 && K >= 1 && K <= STC_MAX_K, STC_ELIMIT, "stc_cheby_dense_fwd_f32: n=%d (<=128) K=%d (<=%d)", n, K, STC_MAX_K);
    STC_REQUIRE(G && T, STC_EINVAL, "stc_cheby_dense_fwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nn = n * n;
    const dim3 grid((nn + EW_THREADS - 1) / EW_THREADS), block(EW_THREADS);
    hipLaunchKernelGGL(cheby_init_kernel, grid, block, 0, s, G, n, K, T);
    for (int k = 2; k < K; ++k)
        hipLaunchKernelGGL(cheby_step_kernel, grid, block, 0, s, G, T + (size_t)(k - 1) * nn, T + (size_t)(k - 2) * nn, T + (size_t)k * nn, n);
    STC_LAUNCH_CHECK("stc_cheby_dense_fwd_f32 launch");
    return STC_OK;
}

extern "C" int stc_cheby_dense_bwd_f32(const float* G, const float* T, float* dT, int32_t n, int32_t K, float* dG, void* stream) {
    STC_REQUIRE(n >= 1 && n <= 128 && K >= 1 && K <= STC_MAX_K, STC_ELIMIT, "stc_cheby_dense_bwd_f32: n=%d (<=128) K=%d (<=%d)", n, K, STC_MAX_K);
    STC_REQUIRE(G && T && dT && dG, STC_EINVAL, "stc_cheby_dense_bwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nn = n * n;
    const dim3 grid((nn + EW_THREADS - 1) / EW_THREADS), block(EW_THREADS);
    if (K < 2) {
        if (int rc = stc::hip_status(hipMemsetAsync(dG, 0, (size_t)nn * sizeof(float), s), "memset dG")) return rc;
        return STC_OK;
    }
    if (K > 2) if (int rc = stc::hip_status(hipMemsetAsync(dG, 0, (size_t)nn * sizeof(float), s), "memset dG")) return rc;
    for (int k = K - 1; k >= 2; --k)
        hipLaunchKernelGGL(cheby_bwd_step_kernel, grid, block, 0, s, G, T + (size_t)(k - 1) * nn, dT + (size_t)k * nn,
                           dT + (size_t)(k - 1) * nn, dT + (size_t)(k - 2) * nn, dG, n);
    hipLaunchKernelGGL(cheby_bwd_final_kernel, grid, block, 0, s, dT + nn, dG, nn, K > 2 ? 1 : 0);
    STC_LAUNCH_CHECK("stc_cheby_dense_bwd_f32 launch");
    return STC_OK;
}

// ---- library-wide ---------------------------------------------------------------
namespace stc {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace stc

extern "C" int stc_version(void) { return STC_ABI_VERSION; }
extern "C" const char* stc_last_error(void) { return stc::error_buffer(); }
