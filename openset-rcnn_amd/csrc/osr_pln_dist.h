// Distance of the Prototype Learning Network between an L2-normalised embedding and an L2-normalised prototype, as
// MODEL.PLN.DISTANCE_TYPE selects it (/root/reference/openset_rcnn/modeling/roi_heads/prototype_learning_network.py:155-160,
// 213-218): 'COS' = 1 - a.b, 'L1' = torch.cdist(p=1), 'L2' = torch.cdist. Both shipped yaml files use 'COS'.
// One wave per (row, prototype) pair, lanes stride the channels.
#pragma once
#include <hip/hip_runtime.h>
#include "osr_common.h"

enum OsrPlnDist { OSR_DIST_COS = 0, OSR_DIST_L1 = 1, OSR_DIST_L2 = 2 };

// ehat(i) returns the i-th component of the first (normalised) vector; p points at the second one.
template <class F>
__device__ __forceinline__ float osr_pln_distance(F ehat, const float* __restrict__ p, int d, int lane, int type) {
    float acc = 0.f;
    if (type == OSR_DIST_COS) {
        for (int i = lane; i < d; i += 64) acc += ehat(i) * p[i];
        return 1.0f - osr_wave_sum(acc);
    }
    if (type == OSR_DIST_L1) {
        for (int i = lane; i < d; i += 64) acc += fabsf(ehat(i) - p[i]);
        return osr_wave_sum(acc);
    }
    for (int i = lane; i < d; i += 64) { const float df = ehat(i) - p[i]; acc += df * df; }
    return sqrtf(osr_wave_sum(acc));
}

// The same with the first vector held in registers: lane l keeps components l, l + 64, ... of it (OSR_PLN_REG * 64 components at
// most), loaded ONCE per row -- the class loop of the PLN kernels then reads only LDS (re-reading the row from global memory
// inside that loop put a memory round trip in front of every one of its K * R distances: 0.2-0.35 ms for a few hundred rows).
#define OSR_PLN_REG 16
// NJ = register slots per lane (NJ * 64 >= d): the kernels are instantiated for 4 (d <= 256: the shipped configs) and OSR_PLN_REG, so the
// class loop of a 256-d embedding runs 4 steps per distance instead of 16 predicated ones.
template <int NJ>
__device__ __forceinline__ float osr_pln_distance_reg(const float (&eh)[NJ], const float* __restrict__ p, int d, int lane, int type) {
    float acc = 0.f;
    if (type == OSR_DIST_COS) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int i = lane + 64 * j; if (i < d) acc += eh[j] * p[i]; }
        return 1.0f - osr_wave_sum(acc);
    }
    if (type == OSR_DIST_L1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int i = lane + 64 * j; if (i < d) acc += fabsf(eh[j] - p[i]); }
        return osr_wave_sum(acc);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int i = lane + 64 * j; if (i < d) { const float df = eh[j] - p[i]; acc += df * df; } }
    return sqrtf(osr_wave_sum(acc));
}

// Four prototypes (consecutive rows of p) at a time: the four partial sums and their four wave reductions are independent, so the
// cross-lane steps of one hide behind the others' (a distance is 6 dependent cross-lane steps; one after the other they were most of a row's time).
// cnt (1..4) of them are valid; the others repeat the first and are ignored by the caller.
template <int NJ>
__device__ __forceinline__ void osr_pln_distance_reg4(const float (&eh)[NJ], const float* __restrict__ p, int d, int cnt, int lane, int type, float (&out)[4]) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* pp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pp[t] = p + (size_t)(t < cnt ? t : 0) * d;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        if (i < d) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float pv = pp[t][i];
                if (type == OSR_DIST_COS) acc[t] += eh[j] * pv;
                else if (type == OSR_DIST_L1) acc[t] += fabsf(eh[j] - pv);
                else { const float df = eh[j] - pv; acc[t] += df * df; }
            }
        }
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += __shfl_xor(acc[t], dd, 64);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = type == OSR_DIST_COS ? 1.0f - acc[t] : (type == OSR_DIST_L1 ? acc[t] : sqrtf(acc[t]));
}

// Row order of the per-row PLN kernels: W waves take the rows W at a time; in pass k wave w takes row k * W + ((w + 131 k) mod W). The
// sampled lists are (image, [foreground ..., background ...]) with a fixed stride that divides W, so a plain w + k W gives a quarter of
// the waves every foreground row (the only rows with work) and the rest none; 131 is odd (a bijection of each pass) and spreads a wave's
// rows over the positions inside a list.
__device__ __forceinline__ long long osr_pln_row(long long k, int w, int W) { return k * W + (int)((w + 131ll * k) % W); }

// d distance(a, b) / d a_i, given the two components and the distance itself (L2 only). d / d b_i: COS -a_i, L1 / L2 the negative.
__device__ __forceinline__ float osr_pln_ddist_da(float a, float b, float dist, int type) {
    if (type == OSR_DIST_COS) return -b;
    const float df = a - b;
    if (type == OSR_DIST_L1) return df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
    return dist > 0.f ? df / dist : 0.f;
}
__device__ __forceinline__ float osr_pln_ddist_db(float a, float b, float dist, int type) {
    if (type == OSR_DIST_COS) return -a;
    return -osr_pln_ddist_da(a, b, dist, type);
}
