// Shared host-side helpers for libstc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "stc_hip.h"

namespace stc {

// thread-local text behind stc_last_error()
char* error_buffer();

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return STC_OK;
    snprintf(error_buffer(), 512, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// MI355X: 256 CUs in 8 XCDs; workgroups are dealt round-robin to the XCDs.
constexpr int kNumXcd = 8;
constexpr int kNumCu = 256;
constexpr int kWave = 64;
constexpr size_t kMaxLdsBytes = 160 * 1024;

// Raise the dynamic-LDS cap of a kernel when it asks for more than the 64 KiB default.
template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Workgroups of `kernel` that are resident at once on the whole device (occupancy API x CU count).
// Persistent kernels size their grid with it: a grid above residency runs in sequential rounds.
template <typename K>
inline int resident_blocks(K kernel, int threads, size_t lds, int fallback_per_cu = 1) {
    int per_cu = 0, dev = 0, cus = kNumCu;
    bool failed = false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) {
        per_cu = fallback_per_cu;
        failed = true;
    }
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else failed = true;
    } else {
        failed = true;
    }
    if (failed) (void)hipGetLastError();      // drop only the error a failed QUERY here left behind, never an earlier launch's
    return per_cu * cus;
}

#define STC_REQUIRE(cond, code, ...) \
    do {                             \
        if (!(cond)) return ::stc::fail(code, __VA_ARGS__); \
    } while (0)

#define STC_LAUNCH_CHECK(what) \
    do {                        \
        hipError_t e__ = hipGetLastError(); \
        if (e__ != hipSuccess) return ::stc::hip_status(e__, what); \
    } while (0)

}  // namespace stc

// ---- device helpers ---------------------------------------------------------
// Contiguous band of tiles per XCD: blocks b and b+8 share an XCD (and its L2),
// so give XCD x the tiles [x*per, (x+1)*per).  Returns -1 for the padding blocks.
__device__ __forceinline__ int stc_xcd_tile(int bid, int n_tiles) {
    const int per = (n_tiles + stc::kNumXcd - 1) / stc::kNumXcd;
    const int tile = (bid % stc::kNumXcd) * per + bid / stc::kNumXcd;
    return tile < n_tiles && (bid / stc::kNumXcd) < per ? tile : -1;
}

// Planes that a launch streams ONCE (an operand plane read by exactly one lane, a result plane nobody re-reads in this launch) carry the
// non-temporal cache policy: on this chip a 2 x 514 MB copy runs at 5.13 TB/s with default-policy accesses, 5.32 with nt stores and 5.63
// with nt loads as well (tools/probes/copy_patterns.hip).  Only for instructions that cover whole 128-byte lines; never on rows that other
// lanes gather again (the aggregation's neighbour rows).
template <class T>
__device__ __forceinline__ T stc_ld_once(const T* p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void stc_st_once(T* p, const T& v) { __builtin_nontemporal_store(v, p); }

// Gate nonlinearities of the fused epilogues on the hardware exp2 / rcp (1 ulp each) -- the IEEE division and libm expf / tanhf are ~30 vector
// instructions each in kernels that are bound by vector issue.  Both are accurate RELATIVE to their result for every argument, as the
// reference's torch.sigmoid / torch.tanh are (STC_GNN.py:72-78): the sigmoid by construction; the tanh form 1 - 2 / (e^{2v} + 1) cancels
// for small arguments (its absolute error of ~1.2e-7 is a relative error of 1.2e-7 / |v|: 1e-3 at |v| = 1e-4), so below 1/4 the odd Taylor
// polynomial through v^9 takes over (truncation 9e-3 v^10 < 1e-8 relative), selected by value -- no branch.  Above 1/4: <= 5e-7 relative.
// (Round 5 probe, HISTORY section 11: libm expf / tanhf, or a compensated exponent argument + a Newton step on the reciprocal, change the
// prediction's error where the model amplifies rounding noise -- graph row sums of 16 .. 50 -- only within its run-to-run spread, 1 - 2.2x the
// reference's own fp32 noise either way; they cost nothing measurable either.  Left as they are.)
__device__ __forceinline__ float stc_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float stc_tanh(float v) {
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * v));      // saturates cleanly at +-1
    const float t = v * v;
    float p = fmaf(t, 62.f / 2835.f, -17.f / 315.f);
    p = fmaf(p, t, 2.f / 15.f);
    p = fmaf(p, t, -1.f / 3.f);
    const float small = fmaf(p * t, v, v);
    return fabsf(v) < 0.25f ? small : big;
}

__device__ __forceinline__ float stc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
