// Shared helpers for the gfx950 kernels behind include/osr.h. Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/osr.h"

#define OSR_WAVE 64

// thread-local last-error message (osr_last_error)
void osr_set_error(const char* fmt, ...);

#define OSR_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            osr_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define OSR_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            osr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return OSR_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) applies to the CURRENT device, whichever host thread asks: one bit per
// device ordinal, set AFTER the attribute call has returned (a second thread that does not see the bit yet repeats the call, which
// is harmless; none can launch on a device whose attribute is still missing).
typedef std::atomic<unsigned long long> osr_dev_mask;
template <class F>
static inline void osr_once_per_device(osr_dev_mask& mask, F&& set_attributes) {
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    const unsigned long long bit = known ? 1ull << dev : 0ull;
    if (known && (mask.load(std::memory_order_acquire) & bit)) return;
    set_attributes();
    if (known) mask.fetch_or(bit, std::memory_order_release);
}

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

__device__ __forceinline__ float osr_to_float(float v) { return v; }
__device__ __forceinline__ float osr_to_float(f16_t v) { return (float)v; }
__device__ __forceinline__ float osr_to_float(bf16_t v) { return (float)v; }

template <class T> __device__ __forceinline__ T osr_from_float(float v);
template <> __device__ __forceinline__ float osr_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ f16_t osr_from_float<f16_t>(float v) { return (f16_t)v; }
template <> __device__ __forceinline__ bf16_t osr_from_float<bf16_t>(float v) { return (bf16_t)v; }

static inline int osr_dtype_size(int dt) { return dt == OSR_F32 ? 4 : 2; }
static inline bool osr_dtype_ok(int dt) { return dt == OSR_F32 || dt == OSR_F16 || dt == OSR_BF16; }

// Order-preserving map fp32 -> uint32 (larger float -> larger key); -0.0 is folded onto +0.0 so that
// equal floats have equal keys (ties are then broken by index, lower first).
__device__ __forceinline__ uint32_t osr_float_key(float f) {
    f = f + 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ bool osr_finite(float v) { return fabsf(v) <= 3.402823466e38f; }  // false for inf/nan

// wave-level inclusive scan of an int (64 lanes)
__device__ __forceinline__ int osr_wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-level exclusive scan (blockDim.x multiple of 64, <= 1024). smem: >= 17 ints. Returns the exclusive
// prefix of v; *total receives the block sum. Contains __syncthreads: call from uniform control flow.
__device__ __forceinline__ int osr_block_excl_scan(int v, int* smem, int* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = osr_wave_incl_scan(v);
    __syncthreads();  // protect smem reuse across calls
    if (lane == 63) smem[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int t = lane < nw ? smem[lane] : 0;
        int ti = osr_wave_incl_scan(t);
        if (lane < nw) smem[lane] = ti - t;
        if (lane == nw - 1) smem[16] = ti;
    }
    __syncthreads();
    int base = smem[wid];
    *total = smem[16];
    return base + inc - v;
}

// One element of SGD with momentum and weight decay (torch.optim.SGD): shared by osr_sgd_step and osr_sgd_step_multi so that the two
// round alike -- contraction into fused multiply-adds is switched off here, otherwise the compiler is free to fuse differently in the
// two kernels and the multi-tensor update would not be bit-identical to the per-tensor one.
__device__ __forceinline__ void osr_sgd_element(float& p, float& v, float g, float rs, float lr, float mu, float wd, float gs) {
#pragma clang fp contract(off)
    const float gi = g * gs * rs + wd * p;
    const float vi = mu * v + gi;
    p = p - lr * vi;
    v = vi;
}

__device__ __forceinline__ float osr_wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
