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
__device__ __forceinline__ float osr_pln_distance_reg(const float (&eh)[OSR_PLN_REG], const float* __restrict__ p, int d, int lane, int type) {
    float acc = 0.f;
    if (type == OSR_DIST_COS) {
#pragma unroll
        for (int j = 0; j < OSR_PLN_REG; ++j) { const int i = lane + 64 * j; if (i < d) acc += eh[j] * p[i]; }
        return 1.0f - osr_wave_sum(acc);
    }
    if (type == OSR_DIST_L1) {
#pragma unroll
        for (int j = 0; j < OSR_PLN_REG; ++j) { const int i = lane + 64 * j; if (i < d) acc += fabsf(eh[j] - p[i]); }
        return osr_wave_sum(acc);
    }
#pragma unroll
    for (int j = 0; j < OSR_PLN_REG; ++j) { const int i = lane + 64 * j; if (i < d) { const float df = eh[j] - p[i]; acc += df * df; } }
    return sqrtf(osr_wave_sum(acc));
}

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
