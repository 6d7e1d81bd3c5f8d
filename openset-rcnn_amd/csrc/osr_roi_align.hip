// RoIAlign over the FPN pyramid for gfx950 (include/osr.h: osr_roi_align_fwd).
//
// Replaces [d2] ROIPooler.forward + torchvision roi_align(aligned=True, sampling_ratio=0) at
// /root/reference/openset_rcnn/modeling/roi_heads/osrcnn_roi_heads.py:108-113,306.
//
// Design (MI355X): one wave per RoI (4 per 256-thread workgroup, no workgroup barriers), NHWC features so that the
// 256 channels of a pixel are one contiguous 512 B (fp16) line read by one wave-instruction (4 channels per lane).
// The adaptive ceil(roi/7) x ceil(roi/7) sample grid of a bin is a tensor product and bilinear weights are products,
// so  sum_samples bilinear(f) = sum_y sum_x wy[y]*wx[x]*f[y][x]  with per-axis weight tables (built by the wave in
// its private LDS slice). Fast path ("column sums"): for each bin row the wave streams the footprint columns left to
// right, forms colsum[x] = sum_y wy[y] f[y][x] from up to 6 row loads (all loads of a column group are issued before
// any is used), and adds wx * colsum into a 3-bin sliding window of register accumulators; a bin is stored as soon
// as the stream has passed it. Every footprint pixel of a bin row is read once instead of 4 taps per sample.
// The validity rule (y<-1||y>H||x<-1||x>W => sample contributes 0) and the edge clamps are per-axis, hence preserved
// exactly; only the fp32 summation order differs from the reference loop (tolerance 1e-4, measured ~1e-6) and the
// final division by the sample count is a multiplication by its reciprocal. RoIs outside the fast path's
// preconditions (bins narrower than a pixel, >6 rows per bin row, >64 footprint columns) take the per-bin separable
// loop, and table overflow (bins wider than 13 px) the per-sample 4-tap loop.
#include "osr_common.h"
#include <stdlib.h>

// Table sizes set the LDS footprint of a wave and with it the occupancy: 4.9 KB per wave -> the kernel is limited by its 85
// VGPRs (5 waves per SIMD) instead of by LDS (4 with the 64 / 352 tables of round 1: 1.51 -> 1.34 ms on the bench's proposals).
#ifndef RA_MAXC
#define RA_MAXC 32  // table columns per bin. Bins of this model's pyramid are at most 7 px wide (28 px RoIs on p2 .. 1333 px on p5);
                    // a wider bin (single-level pyramids in the tests) takes the per-sample loop
#endif
#ifndef RA_MAXX
#define RA_MAXX 192 // steps of the streamed (shorter) side of the footprint on the column-sum path; beyond: the per-bin loop
#endif
#ifndef RA_DEPTH
#define RA_DEPTH 2  // register sets of the software pipeline of ra_bin_row (4 measured: no gain, the kernel is not latency-bound)
#endif

struct RoiAlignArgs {
    const void* data[4];
    int h[4], w[4];
    float scale[4];
    int num_levels, c;
    const float* boxes;
    const int* batch_idx;
    long long m;
    int pooled, canonical_level, canonical_size, min_level;
    void* out;
};

template <class T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<f16_t> { typedef uint2 type; };
template <> struct Vec4<bf16_t> { typedef uint2 type; };

template <class T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float v[4]) {
    typedef f16_t h4 __attribute__((ext_vector_type(4)));
    h4 t = *reinterpret_cast<const h4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <class T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float v[4]) {
    typedef f16_t h4 __attribute__((ext_vector_type(4)));
    h4 t = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
    *reinterpret_cast<h4*>(p) = t;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
    typedef bf16_t b4 __attribute__((ext_vector_type(4)));
    b4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<b4*>(p) = t;
}

template <class T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) { load4<float>(p, v); load4<float>(p + 4, v + 4); }
template <> __device__ __forceinline__ void load8<f16_t>(const f16_t* p, float v[8]) {
    typedef f16_t h8 __attribute__((ext_vector_type(8)));
    h8 t = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float v[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}
// packed 8-channel register image of one load (kept packed while the loads of a column group are in flight)
typedef unsigned int ra_u32x4 __attribute__((ext_vector_type(4)));
typedef float ra_f32x8 __attribute__((ext_vector_type(8)));
template <class T> struct Raw8;
template <> struct Raw8<float> {
    ra_f32x8 r;
    __device__ __forceinline__ void load(const float* p) { r = *reinterpret_cast<const ra_f32x8*>(p); }
    __device__ __forceinline__ void get(float v[8]) const { _Pragma("unroll") for (int i = 0; i < 8; ++i) v[i] = r[i]; }
};
template <> struct Raw8<f16_t> {
    typedef f16_t h8 __attribute__((ext_vector_type(8)));
    h8 r;
    __device__ __forceinline__ void load(const f16_t* p) { r = *reinterpret_cast<const h8*>(p); }
    __device__ __forceinline__ void get(float v[8]) const { _Pragma("unroll") for (int i = 0; i < 8; ++i) v[i] = (float)r[i]; }
};
template <> struct Raw8<bf16_t> {
    ra_u32x4 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const ra_u32x4*>(p); }
    __device__ __forceinline__ void get(float v[8]) const {
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
    }
};
// packed 4-channel register image of one 8-byte (fp16/bf16) or 16-byte (fp32) load
template <class T> struct Raw4;
template <> struct Raw4<float> {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 r;
    __device__ __forceinline__ void load(const float* p) { r = *reinterpret_cast<const f4*>(p); }
    __device__ __forceinline__ void get(float v[4]) const { _Pragma("unroll") for (int i = 0; i < 4; ++i) v[i] = r[i]; }
};
template <> struct Raw4<f16_t> {
    typedef f16_t h4 __attribute__((ext_vector_type(4)));
    h4 r;
    __device__ __forceinline__ void load(const f16_t* p) { r = *reinterpret_cast<const h4*>(p); }
    __device__ __forceinline__ void get(float v[4]) const { _Pragma("unroll") for (int i = 0; i < 4; ++i) v[i] = (float)r[i]; }
};
template <> struct Raw4<bf16_t> {
    typedef bf16_t b4 __attribute__((ext_vector_type(4)));
    b4 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const b4*>(p); }
    __device__ __forceinline__ void get(float v[4]) const { _Pragma("unroll") for (int i = 0; i < 4; ++i) v[i] = (float)r[i]; }
};
template <class T> __device__ __forceinline__ void store8(T* p, const float v[8]) { store4<T>(p, v); store4<T>(p + 4, v + 4); }
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float v[8]) {
    typedef f16_t h8 __attribute__((ext_vector_type(8)));
    h8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (f16_t)v[i];
    *reinterpret_cast<h8*>(p) = t;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float v[8]) {
    typedef bf16_t b8 __attribute__((ext_vector_type(8)));
    b8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<b8*>(p) = t;
}

// One sample coordinate of torchvision's pre_calc_for_bilinear_interpolate along one axis.
// Returns false when the sample is outside [-1, size] (contributes nothing).
__device__ __forceinline__ bool axis_sample(float start, int bin, float bin_size, int i, int grid, int size,
                                            int* lo, int* hi, float* wl, float* wh) {
    float v = start + bin * bin_size + ((float)i + .5f) * bin_size / (float)grid;
    if (v < -1.0f || v > (float)size) return false;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= size - 1) { h = l = size - 1; v = (float)l; } else h = l + 1;
    float f = v - (float)l;
    *lo = l; *hi = h; *wh = f; *wl = 1.f - f;
    return true;
}

__device__ __forceinline__ void ra_wave_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();     // and the compiler keeps later LDS reads behind them
}

struct RaWaveLds {
    float w[2][7][RA_MAXC];  // [axis: 0 = y, 1 = x][bin][column of the bin's footprint]
    int lo[2][8], n[2][8];
    int colb[RA_MAXX];       // per footprint column: first unfinished bin
    float colw[3][RA_MAXX];  // weight of the column in bins colb, colb+1, colb+2
};

#ifndef RA_WPB
#define RA_WPB 2  // waves (= RoIs) per workgroup (2: -3 % against 4 once the small tables let five waves per SIMD in)
#endif
#ifndef RA_PG
#define RA_PG 1    // columns per pipelined step
#endif

// One bin of the inner axis of one RoI for this lane's 4 channels, streamed along the outer axis, software pipelined: while the RA_PG steps of one
// step are reduced, the loads of the next step are already in flight (two register sets, used alternately). NY (rows of the
// bin row's footprint, wave-uniform) is a template parameter so that every step issues the same number of loads and the
// compiler can place counted waits; columns past the footprint re-read its last column (a cache hit) instead of branching.
// Same arithmetic, in the same order, as the un-pipelined loop.
template <int NY, class TI, class TO>
__device__ __forceinline__ void ra_bin_row(const TI* __restrict__ rp, size_t rowstride, size_t sstride, int ncol, const RaWaveLds& S,
                                           const float (&wy)[6], float inv_count, TO* __restrict__ outrow, size_t ostride, bool cok, int P) {
    float a0[4], a1[4], a2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a0[k] = 0.f; a1[k] = 0.f; a2[k] = 0.f; }
    int bcur = 0;
    Raw4<TI> va[RA_PG][NY], vb[RA_PG][NY];
#define RA_ISSUE(v, xg_)                                                                                   \
    {                                                                                                      \
        _Pragma("unroll") for (int g2 = 0; g2 < RA_PG; ++g2) {                                             \
            const int col_ = (xg_) + g2 < ncol ? (xg_) + g2 : ncol - 1;                                    \
            const TI* cp_ = rp + (size_t)col_ * sstride;                                                   \
            _Pragma("unroll") for (int j = 0; j < NY; ++j) v[g2][j].load(cp_ + j * rowstride);             \
        }                                                                                                  \
    }
#define RA_FLUSH2()                                                                                        \
    {                                                                                                      \
        float tot[4];                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                    \
            tot[k] = a0[k] * inv_count;                                                                    \
            a0[k] = a1[k]; a1[k] = a2[k]; a2[k] = 0.f;                                                     \
        }                                                                                                  \
        if (cok) store4<TO>(outrow + (size_t)bcur * ostride, tot);                                         \
        ++bcur;                                                                                            \
    }
#define RA_CONSUME(v, xg_)                                                                                 \
    {                                                                                                      \
        _Pragma("unroll") for (int g2 = 0; g2 < RA_PG; ++g2) {                                             \
            if ((xg_) + g2 < ncol) {                                                                       \
                const int x = (xg_) + g2;                                                                  \
                const int cbx = __builtin_amdgcn_readfirstlane(S.colb[x]);                                 \
                while (bcur < cbx) RA_FLUSH2();                                                            \
                float cs[4] = {0.f, 0.f, 0.f, 0.f}, f[4];                                                  \
                _Pragma("unroll") for (int j = 0; j < NY; ++j) {                                           \
                    v[g2][j].get(f);                                                                       \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) cs[k] = __builtin_fmaf(wy[j], f[k], cs[k]); \
                }                                                                                          \
                const float w0 = S.colw[0][x], w1 = S.colw[1][x], w2 = S.colw[2][x];                       \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
                    a0[k] = __builtin_fmaf(w0, cs[k], a0[k]);                                              \
                    a1[k] = __builtin_fmaf(w1, cs[k], a1[k]);                                              \
                    a2[k] = __builtin_fmaf(w2, cs[k], a2[k]);                                              \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
    }
#if RA_DEPTH == 4
    Raw4<TI> vc[RA_PG][NY], vd[RA_PG][NY];
    RA_ISSUE(va, 0);
    RA_ISSUE(vb, RA_PG);
    RA_ISSUE(vc, 2 * RA_PG);
    for (int xg = 0; xg < ncol; xg += 4 * RA_PG) {
        RA_ISSUE(vd, xg + 3 * RA_PG);
        RA_CONSUME(va, xg);
        RA_ISSUE(va, xg + 4 * RA_PG);
        RA_CONSUME(vb, xg + RA_PG);
        RA_ISSUE(vb, xg + 5 * RA_PG);
        RA_CONSUME(vc, xg + 2 * RA_PG);
        RA_ISSUE(vc, xg + 6 * RA_PG);
        RA_CONSUME(vd, xg + 3 * RA_PG);
    }
#else
    RA_ISSUE(va, 0);
    for (int xg = 0; xg < ncol; xg += 2 * RA_PG) {
        RA_ISSUE(vb, xg + RA_PG);
        RA_CONSUME(va, xg);
        RA_ISSUE(va, xg + 2 * RA_PG);
        RA_CONSUME(vb, xg + RA_PG);
    }
#endif
    while (bcur < P) RA_FLUSH2();
#undef RA_ISSUE
#undef RA_CONSUME
#undef RA_FLUSH2
}

// The same bin row when its footprint is taller than 6 feature rows (tall boxes: up to H / 7 + 2 rows per bin): the column sum
// runs over the rows in chunks of 6 loads in flight, row weights come from the LDS table.
template <class TI, class TO>
__device__ __forceinline__ void ra_bin_row_tall(const TI* __restrict__ rp, size_t rowstride, size_t sstride, int ncol, const RaWaveLds& S,
                                                const float* wrow, int ny, float inv_count, TO* __restrict__ outrow, size_t ostride, bool cok, int P) {
    float a0[4], a1[4], a2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a0[k] = 0.f; a1[k] = 0.f; a2[k] = 0.f; }
    int bcur = 0;
    for (int x = 0; x < ncol; ++x) {
        const int cbx = __builtin_amdgcn_readfirstlane(S.colb[x]);
        while (bcur < cbx) {
            float tot[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { tot[k] = a0[k] * inv_count; a0[k] = a1[k]; a1[k] = a2[k]; a2[k] = 0.f; }
            if (cok) store4<TO>(outrow + (size_t)bcur * ostride, tot);
            ++bcur;
        }
        const TI* cp = rp + (size_t)x * sstride;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < ny; j0 += 6) {
            Raw4<TI> v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j0 + j < ny) v[j].load(cp + (size_t)(j0 + j) * rowstride);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j0 + j < ny) {
                    float f[4];
                    v[j].get(f);
                    const float wj = wrow[j0 + j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) cs[k] = __builtin_fmaf(wj, f[k], cs[k]);
                }
        }
        const float w0 = S.colw[0][x], w1 = S.colw[1][x], w2 = S.colw[2][x];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a0[k] = __builtin_fmaf(w0, cs[k], a0[k]);
            a1[k] = __builtin_fmaf(w1, cs[k], a1[k]);
            a2[k] = __builtin_fmaf(w2, cs[k], a2[k]);
        }
    }
    while (bcur < P) {
        float tot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { tot[k] = a0[k] * inv_count; a0[k] = a1[k]; a1[k] = a2[k]; a2[k] = 0.f; }
        if (cok) store4<TO>(outrow + (size_t)bcur * ostride, tot);
        ++bcur;
    }
}

// One wave per RoI (RA_WPB RoIs per workgroup, no workgroup barriers). The wave builds the per-axis weight tables in its
// private LDS slice, picks the shorter side of the footprint as the streamed axis, and then, for each of the 7 bins of the
// other axis, walks the footprint one pixel column (or row) at a time: the pixels of the step that fall into the bin are
// reduced with the bin's weights (software pipelined, the next step's loads in flight) and the sum goes into a 3-bin
// sliding window of register accumulators along the streamed axis.
#ifndef RA_MINW
#define RA_MINW 1  // waves per SIMD the register allocation must allow (occupancy is otherwise limited by the LDS tables)
#endif
template <class TI, class TO>
__global__ __launch_bounds__(RA_WPB * 64, RA_MINW) void roi_align_kernel(RoiAlignArgs a) {
    __shared__ RaWaveLds s_all[RA_WPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: workgroup b runs on XCD b % 8, so each XCD walks one contiguous eighth of the RoI list and RoIs that are
    // neighbours in the list (and, when the list is spatially ordered, in the image) share an L2
    long long r;
    {
        const int nwg = gridDim.x, bq = blockIdx.x, q = nwg >> 3, rr = nwg & 7, xcd = bq & 7, idx = bq >> 3;
        const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
        r = (long long)t * RA_WPB + wid;
    }
    if (r >= a.m) return;
    RaWaveLds& S = s_all[wid];
    const int P = a.pooled, C = a.c;
    TO* out = reinterpret_cast<TO*>(a.out) + (size_t)r * P * P * C;

    const int b = a.batch_idx[r];
    if (b < 0) {  // padding row: zeros
        for (int i = lane * 4; i < P * P * C; i += 64 * 4) {
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            store4<TO>(out + i, z);
        }
        return;
    }
    const float bx1 = a.boxes[r * 4 + 0], by1 = a.boxes[r * 4 + 1], bx2 = a.boxes[r * 4 + 2], by2 = a.boxes[r * 4 + 3];
    // [d2] assign_boxes_to_levels, evaluated in fp32 exactly as written there
    float sz = sqrtf((bx2 - bx1) * (by2 - by1));
    float lvf = floorf((float)a.canonical_level + log2f(sz / (float)a.canonical_size + 1e-8f));
    float lmin = (float)a.min_level, lmax = (float)(a.min_level + a.num_levels - 1);
    lvf = fminf(fmaxf(lvf, lmin), lmax);  // NaN (degenerate area) -> lmin via fmaxf
    const int lv = __builtin_amdgcn_readfirstlane((int)lvf - a.min_level);
    const int H = a.h[lv], W = a.w[lv];
    const float scale = a.scale[lv];
    const TI* feat = reinterpret_cast<const TI*>(a.data[lv]) + (size_t)b * H * W * C;

    const float sw = bx1 * scale - 0.5f, sh = by1 * scale - 0.5f;
    const float ew = bx2 * scale - 0.5f, eh = by2 * scale - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);

    // ---- per-axis weight tables: entry (axis, bin, col) sums the samples that touch its column ----
    bool overflow = false;
    // (a bin's samples span its width + 1 pixels, so only the first max(gh, gw) + 3 table columns can be non-zero and only
    // those are ever read: build just them)
    const int tcols = min(RA_MAXC, max(max(gh, gw), 1) + 3);
    for (int e = lane; e < 2 * 7 * tcols; e += 64) {
        const int axis = e / (7 * tcols), bin = (e / tcols) % 7, col = e % tcols;
        if (bin >= P) continue;
        const float start = axis ? sw : sh, bs = axis ? bw : bh;
        const int grid = axis ? gw : gh, size = axis ? W : H;
        int first = -1, last = -1;
        float acc = 0.f;
        for (int i = 0; i < grid; ++i) {
            int lo, hi; float wl, wh;
            if (!axis_sample(start, bin, bs, i, grid, size, &lo, &hi, &wl, &wh)) continue;
            if (first < 0) first = lo;
            last = hi;
            if (lo - first == col) acc += wl;
            if (hi - first == col) acc += wh;
        }
        S.w[axis][bin][col] = acc;
        if (col == 0) {
            const int n = first < 0 ? 0 : last - first + 1;
            S.lo[axis][bin] = first < 0 ? 0 : first;
            S.n[axis][bin] = n;
            overflow |= n > tcols;
        }
    }
    const bool fallback = __any(overflow);
    ra_wave_sync();

    // ---- streaming fast path. The footprint is walked along one axis (the "outer" axis, one step per pixel column or row);
    //      per step the pixels of the other ("inner") axis that fall into the current bin are reduced with the bin's weights,
    //      and the result is scattered into a 3-bin sliding window of accumulators along the outer axis. The outer axis is
    //      the SHORTER side of the footprint: a step costs ~35 instructions however few pixels it reduces, and the proposals
    //      are 2-10x wider than tall (or the reverse) often enough that streaming the long side doubled the kernel's time.
    //      Preconditions (checked by lanes 0..P-1, one bin each, for both axes): bins ordered, no holes, no pixel in more
    //      than three consecutive bins. ----
    bool bad[2] = {fallback, fallback};
    int lo_l[2] = {0x7fffffff, 0x7fffffff}, hi_l[2] = {0, 0};
    if (lane < P) {
#pragma unroll
        for (int ax = 0; ax < 2; ++ax) {
            const int lo = S.lo[ax][lane], n = S.n[ax][lane];
            if (n > 0) { lo_l[ax] = lo; hi_l[ax] = lo + n; }
            if (lane + 1 < P && n > 0 && S.n[ax][lane + 1] > 0 && S.lo[ax][lane + 1] < lo) bad[ax] = true;
            if (lane + 3 < P && n > 0 && S.n[ax][lane + 3] > 0 && S.lo[ax][lane + 3] < lo + n) bad[ax] = true;
            if (lane + 1 < P && lane > 0 && n == 0 && S.n[ax][lane - 1] > 0 && S.n[ax][lane + 1] > 0) bad[ax] = true;  // hole: not expected
        }
    }
#pragma unroll
    for (int ax = 0; ax < 2; ++ax)
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            lo_l[ax] = min(lo_l[ax], __shfl_xor(lo_l[ax], d, 64));
            hi_l[ax] = max(hi_l[ax], __shfl_xor(hi_l[ax], d, 64));
        }
    int ext_lo[2], ext_n[2];
    bool ax_ok[2];
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
        const int l = __builtin_amdgcn_readfirstlane(lo_l[ax]), h = __builtin_amdgcn_readfirstlane(hi_l[ax]);
        ext_lo[ax] = l == 0x7fffffff ? 0 : l;
        ext_n[ax] = l == 0x7fffffff ? 0 : h - l;
        ax_ok[ax] = !__any(bad[ax]) && ext_n[ax] <= RA_MAXX;
    }
    // outer (streamed) axis: the shorter side when its preconditions hold, else the other one
#ifndef RA_AXIS_SELECT
#define RA_AXIS_SELECT 1
#endif
    int oa = (!RA_AXIS_SELECT || ext_n[1] <= ext_n[0]) ? 1 : 0;
    if (!ax_ok[oa]) oa ^= 1;

    if (ax_ok[oa]) {
        const int ia = oa ^ 1;
        const int os = ext_lo[oa], nstep = ext_n[oa];
        for (int sl = lane; sl < nstep; sl += 64) {  // per-step table: first unfinished bin of this outer pixel + its weights in 3 bins
            const int x = os + sl;
            int cb = 0;
            while (cb < P && (S.n[oa][cb] == 0 || x >= S.lo[oa][cb] + S.n[oa][cb])) ++cb;
            S.colb[sl] = cb;
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2) {
                const int bb = cb + t2;
                float wv = 0.f;
                if (bb < P) { const int i = x - S.lo[oa][bb]; if (i >= 0 && i < S.n[oa][bb]) wv = S.w[oa][bb][i]; }
                S.colw[t2][sl] = wv;
            }
        }
        ra_wave_sync();
        const float inv_count = 1.0f / count;
        const size_t rowstride = (size_t)W * C;
        // element strides of one step along the inner / outer axis, and between two consecutive output bins of the outer axis
        const size_t istride = ia == 0 ? rowstride : (size_t)C, sstride = ia == 0 ? (size_t)C : rowstride;
        const size_t ostride = oa == 1 ? (size_t)C : (size_t)P * C;
        for (int pb = 0; pb < P; ++pb) {  // bins along the inner axis
            const int i0 = __builtin_amdgcn_readfirstlane(S.lo[ia][pb]), ni = __builtin_amdgcn_readfirstlane(S.n[ia][pb]);
            float wi[6];  // wave-uniform inner weights (live in scalar registers)
#pragma unroll
            for (int j = 0; j < 6; ++j) wi[j] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(j < ni ? S.w[ia][pb][j] : 0.f)));
            for (int cb0 = 0; cb0 < C; cb0 += 256) {
                const int c0 = cb0 + lane * 4;
                const bool cok = c0 < C;
                const TI* rp = feat + (size_t)i0 * istride + (size_t)os * sstride + (cok ? c0 : 0);
                TO* outrow = out + (size_t)pb * (oa == 1 ? (size_t)P * C : (size_t)C) + c0;
                switch (nstep > 0 ? (ni > 6 ? 7 : ni) : 0) {  // (no step: the pipelined loop would have nothing valid to prefetch)
                    case 1: ra_bin_row<1, TI, TO>(rp, istride, sstride, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 2: ra_bin_row<2, TI, TO>(rp, istride, sstride, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 3: ra_bin_row<3, TI, TO>(rp, istride, sstride, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 4: ra_bin_row<4, TI, TO>(rp, istride, sstride, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 5: ra_bin_row<5, TI, TO>(rp, istride, sstride, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 6: ra_bin_row<6, TI, TO>(rp, istride, sstride, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 7: ra_bin_row_tall<TI, TO>(rp, istride, sstride, nstep, S, S.w[ia][pb], ni, inv_count, outrow, ostride, cok, P); break;
                    default: {  // no valid sample in this bin row / column: zeros
                        const float z[4] = {0.f, 0.f, 0.f, 0.f};
                        if (cok) for (int q = 0; q < P; ++q) store4<TO>(outrow + (size_t)q * ostride, z);
                    }
                }
            }
        }
        return;
    }

    // ---- general paths: per-bin separable footprint, or (table overflow) the per-sample 4-tap loop ----
    for (int ph = 0; ph < P; ++ph) {
        for (int c0 = lane * 4; c0 < C; c0 += 256) {
            for (int pw = 0; pw < P; ++pw) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                if (!fallback) {
                    const int y0 = S.lo[0][ph], ny = S.n[0][ph], x0 = S.lo[1][pw], nx = S.n[1][pw];
                    for (int j = 0; j < ny; ++j) {
                        const float wy = S.w[0][ph][j];
                        const TI* row = feat + ((size_t)(y0 + j) * W + x0) * C + c0;
                        for (int i = 0; i < nx; ++i) {
                            float v[4];
                            load4<TI>(row + (size_t)i * C, v);
                            const float wgt = wy * S.w[1][pw][i];
                            acc[0] += wgt * v[0]; acc[1] += wgt * v[1]; acc[2] += wgt * v[2]; acc[3] += wgt * v[3];
                        }
                    }
                } else {
                    for (int iy = 0; iy < gh; ++iy) {
                        int yl, yh; float hy, ly;
                        if (!axis_sample(sh, ph, bh, iy, gh, H, &yl, &yh, &hy, &ly)) continue;
                        for (int ix = 0; ix < gw; ++ix) {
                            int xl, xh; float hx, lx;
                            if (!axis_sample(sw, pw, bw, ix, gw, W, &xl, &xh, &hx, &lx)) continue;
                            float v1[4], v2[4], v3[4], v4[4];
                            load4<TI>(feat + ((size_t)yl * W + xl) * C + c0, v1);
                            load4<TI>(feat + ((size_t)yl * W + xh) * C + c0, v2);
                            load4<TI>(feat + ((size_t)yh * W + xl) * C + c0, v3);
                            load4<TI>(feat + ((size_t)yh * W + xh) * C + c0, v4);
                            const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[k] += w1 * v1[k] + w2 * v2[k] + w3 * v3[k] + w4 * v4[k];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = acc[k] / count;
                store4<TO>(out + (size_t)(ph * P + pw) * C + c0, acc);
            }
        }
    }
}

template <class TI>
static osr_status launch_out(const RoiAlignArgs& a, int out_dtype, hipStream_t st) {
    dim3 grid((unsigned)((a.m + RA_WPB - 1) / RA_WPB)), block(RA_WPB * 64);
    switch (out_dtype) {
        case OSR_F32: hipLaunchKernelGGL((roi_align_kernel<TI, float>), grid, block, 0, st, a); break;
        case OSR_F16: hipLaunchKernelGGL((roi_align_kernel<TI, f16_t>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((roi_align_kernel<TI, bf16_t>), grid, block, 0, st, a); break;
    }
    OSR_CHECK_LAUNCH("osr_roi_align_fwd");
    return OSR_OK;
}

extern "C" osr_status osr_roi_align_fwd(const osr_pyramid* f, int32_t feat_dtype, int32_t n, const float* boxes,
                                        const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                        int32_t canonical_size, int32_t min_level, void* out, int32_t out_dtype,
                                        void* stream) {
    OSR_REQUIRE(f && boxes && batch_idx && out, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd: null pointer");
    OSR_REQUIRE(f->num_levels >= 1 && f->num_levels <= 4, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd: 1..4 levels, got %d", f->num_levels);
    OSR_REQUIRE(pooled >= 1 && pooled <= 7, OSR_ERR_UNSUPPORTED, "osr_roi_align_fwd: pooled size 1..7, got %d", pooled);
    OSR_REQUIRE(f->c > 0 && f->c % 4 == 0, OSR_ERR_UNSUPPORTED, "osr_roi_align_fwd: channels must be a multiple of 4, got %d", f->c);
    OSR_REQUIRE(osr_dtype_ok(feat_dtype) && osr_dtype_ok(out_dtype), OSR_ERR_INVALID_ARG, "osr_roi_align_fwd: bad dtype");
    OSR_REQUIRE(n >= 1 && m >= 0 && m < (1ll << 31), OSR_ERR_INVALID_ARG, "osr_roi_align_fwd: bad n/m");
    OSR_REQUIRE(canonical_size > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd: canonical_size must be > 0");
    if (m == 0) return OSR_OK;
    RoiAlignArgs a;
    for (int l = 0; l < 4; ++l) {
        int s = l < f->num_levels ? l : 0;
        OSR_REQUIRE(f->data[s] && f->h[s] > 0 && f->w[s] > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd: bad level %d", s);
        a.data[l] = f->data[s]; a.h[l] = f->h[s]; a.w[l] = f->w[s]; a.scale[l] = f->scale[s];
    }
    a.num_levels = f->num_levels; a.c = f->c; a.boxes = boxes; a.batch_idx = batch_idx; a.m = m;
    a.pooled = pooled; a.canonical_level = canonical_level; a.canonical_size = canonical_size; a.min_level = min_level;
    a.out = out;
    hipStream_t st = (hipStream_t)stream;
    switch (feat_dtype) {
        case OSR_F32: return launch_out<float>(a, out_dtype, st);
        case OSR_F16: return launch_out<f16_t>(a, out_dtype, st);
        default: return launch_out<bf16_t>(a, out_dtype, st);
    }
}

// ------------------------------------------------------------------------------------------------------
// RoIAlign backward: d feature pyramid (fp32, zero-initialised by the caller) += scatter of d out.
// Same geometry, level assignment and per-axis weight tables as the forward kernel: the gradient of a bin spreads over
// its footprint pixels with weight wy[j] * wx[i] / count (per-sample 4-tap scatter when the tables overflow). One wave per
// RoI, 4 channels per lane, fp32 atomic adds (several RoIs overlap on the same pixels; the summation order, and with it the
// last bits of the result, therefore vary from run to run -- the reference's atomicAdd backward does the same).
// ------------------------------------------------------------------------------------------------------
struct RoiAlignBwdArgs {
    float* data[4];
    int h[4], w[4];
    float scale[4];
    int num_levels, c;
    const float* boxes;
    const int* batch_idx;
    long long m;
    int pooled, canonical_level, canonical_size, min_level;
    const void* dout;
};

#define RA_BWD_WPB 2   // RoIs per workgroup of the backward kernel (20 KB of LDS per RoI)
#define RA_MAXY 208    // rows of the whole RoI footprint on the streaming path (p2 of an 800 px high batch: 200)
struct RaBwdLds {
    RaWaveLds t;                 // per-axis bin tables + per-column (x) window table
    int rowb[RA_MAXY];           // per footprint row: first unfinished bin
    float roww[3][RA_MAXY];      // weight of the row in bins rowb, rowb+1, rowb+2
    float tb[7][4][64];          // per lane: the row's gradient folded over y, for each x bin and each of the lane's 4 channels
};

template <class TG>
__global__ __launch_bounds__(RA_BWD_WPB * 64) void roi_align_bwd_kernel(RoiAlignBwdArgs a) {
    __shared__ RaBwdLds s_all[RA_BWD_WPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: workgroup b runs on XCD b % 8, so each XCD walks one contiguous eighth of the RoI list and RoIs that are
    // neighbours in the list (and, when the list is spatially ordered, in the image) share an L2
    long long r;
    {
        const int nwg = gridDim.x, bq = blockIdx.x, q = nwg >> 3, rr = nwg & 7, xcd = bq & 7, idx = bq >> 3;
        const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
        r = (long long)t * RA_BWD_WPB + wid;
    }
    if (r >= a.m) return;
    RaBwdLds& SB = s_all[wid];
    RaWaveLds& S = SB.t;
    const int P = a.pooled, C = a.c;
    const TG* dout = reinterpret_cast<const TG*>(a.dout) + (size_t)r * P * P * C;
    const int b = a.batch_idx[r];
    if (b < 0) return;
    const float bx1 = a.boxes[r * 4 + 0], by1 = a.boxes[r * 4 + 1], bx2 = a.boxes[r * 4 + 2], by2 = a.boxes[r * 4 + 3];
    float sz = sqrtf((bx2 - bx1) * (by2 - by1));
    float lvf = floorf((float)a.canonical_level + log2f(sz / (float)a.canonical_size + 1e-8f));
    float lmin = (float)a.min_level, lmax = (float)(a.min_level + a.num_levels - 1);
    lvf = fminf(fmaxf(lvf, lmin), lmax);
    const int lv = __builtin_amdgcn_readfirstlane((int)lvf - a.min_level);
    const int H = a.h[lv], W = a.w[lv];
    const float scale = a.scale[lv];
    float* feat = a.data[lv] + (size_t)b * H * W * C;
    const float sw = bx1 * scale - 0.5f, sh = by1 * scale - 0.5f;
    const float ew = bx2 * scale - 0.5f, eh = by2 * scale - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    bool overflow = false;
    // (a bin's samples span its width + 1 pixels, so only the first max(gh, gw) + 3 table columns can be non-zero and only
    // those are ever read: build just them)
    const int tcols = min(RA_MAXC, max(max(gh, gw), 1) + 3);
    for (int e = lane; e < 2 * 7 * tcols; e += 64) {
        const int axis = e / (7 * tcols), bin = (e / tcols) % 7, col = e % tcols;
        if (bin >= P) continue;
        const float start = axis ? sw : sh, bs = axis ? bw : bh;
        const int grid = axis ? gw : gh, size = axis ? W : H;
        int first = -1, last = -1;
        float acc = 0.f;
        for (int i = 0; i < grid; ++i) {
            int lo, hi; float wl, wh;
            if (!axis_sample(start, bin, bs, i, grid, size, &lo, &hi, &wl, &wh)) continue;
            if (first < 0) first = lo;
            last = hi;
            if (lo - first == col) acc += wl;
            if (hi - first == col) acc += wh;
        }
        S.w[axis][bin][col] = acc;
        if (col == 0) {
            const int n = first < 0 ? 0 : last - first + 1;
            S.lo[axis][bin] = first < 0 ? 0 : first;
            S.n[axis][bin] = n;
            overflow |= n > tcols;
        }
    }
    const bool fallback = __any(overflow);
    ra_wave_sync();
    // ---- streaming path: ONE atomic per footprint pixel and channel. d feat[y][x] = sum_by sum_bx wy[by][y] wx[bx][x] g[by][bx]
    //      is evaluated row by row: the three bin rows a pixel row can belong to are held in registers (gwin, sliding down the
    //      RoI), folded over y into t[bx] (7 values per channel, parked in the lane's LDS column), and a 3-bin window of t slides
    //      along x. The per-bin loop below it issues ny*nx atomics per bin instead (3-4x more) and remains the fallback when a
    //      pixel touches more than three bins of an axis or the tables overflow. ----
    {
        bool bad = fallback;
        int lo_l[2] = {0x7fffffff, 0x7fffffff}, hi_l[2] = {0, 0};
        if (lane < P) {
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                const int lo = S.lo[ax][lane], n = S.n[ax][lane];
                if (n > 0) { lo_l[ax] = lo; hi_l[ax] = lo + n; }
                if (lane + 1 < P && n > 0 && S.n[ax][lane + 1] > 0 && S.lo[ax][lane + 1] < lo) bad = true;
                if (lane + 3 < P && n > 0 && S.n[ax][lane + 3] > 0 && S.lo[ax][lane + 3] < lo + n) bad = true;
                if (lane + 1 < P && lane > 0 && n == 0 && S.n[ax][lane - 1] > 0 && S.n[ax][lane + 1] > 0) bad = true;
            }
        }
#pragma unroll
        for (int ax = 0; ax < 2; ++ax)
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                lo_l[ax] = min(lo_l[ax], __shfl_xor(lo_l[ax], d, 64));
                hi_l[ax] = max(hi_l[ax], __shfl_xor(hi_l[ax], d, 64));
            }
        const int ys0 = __builtin_amdgcn_readfirstlane(lo_l[0]), ye0 = __builtin_amdgcn_readfirstlane(hi_l[0]);
        const int xs0 = __builtin_amdgcn_readfirstlane(lo_l[1]), xe0 = __builtin_amdgcn_readfirstlane(hi_l[1]);
        const bool empty = ys0 == 0x7fffffff || xs0 == 0x7fffffff;
        if (empty && !__any(bad)) return;  // no valid sample: no gradient
        const int nrow = empty ? 0 : ye0 - ys0, ncol = empty ? 0 : xe0 - xs0;
        if (!__any(bad) && nrow <= RA_MAXY && ncol <= RA_MAXX) {
            for (int sl = lane; sl < nrow + ncol; sl += 64) {  // per-pixel window tables of both axes
                const bool isx = sl >= nrow;
                const int ax = isx ? 1 : 0, i0 = isx ? sl - nrow : sl, x = (isx ? xs0 : ys0) + i0;
                int cb = 0;
                while (cb < P && (S.n[ax][cb] == 0 || x >= S.lo[ax][cb] + S.n[ax][cb])) ++cb;
                float wv[3];
#pragma unroll
                for (int t2 = 0; t2 < 3; ++t2) {
                    const int bb = cb + t2;
                    wv[t2] = 0.f;
                    if (bb < P) { const int i = x - S.lo[ax][bb]; if (i >= 0 && i < S.n[ax][bb]) wv[t2] = S.w[ax][bb][i]; }
                }
                if (isx) { S.colb[i0] = cb; S.colw[0][i0] = wv[0]; S.colw[1][i0] = wv[1]; S.colw[2][i0] = wv[2]; }
                else { SB.rowb[i0] = cb; SB.roww[0][i0] = wv[0]; SB.roww[1][i0] = wv[1]; SB.roww[2][i0] = wv[2]; }
            }
            ra_wave_sync();
            const float inv_count = 1.0f / count;
            for (int cb0 = 0; cb0 < C; cb0 += 256) {
                bool chok[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) chok[k] = cb0 + k * 64 + lane < C;
                // gwin[d][bx][k]: upstream gradient of bin row (rcur + d), bin column bx, channel cb0 + k*64 + lane (already / count)
                float gwin[3][7][4];
#define RA_G_AT(byrow, j, k) (((byrow) < P && (j) < P && chok[k]) ? osr_to_float(dout[(size_t)((byrow) * P + (j)) * C + cb0 + (k) * 64 + lane]) * inv_count : 0.f)
                int rcur = -1;  // bin row held in gwin[0]; -1: nothing loaded yet
                for (int yi = 0; yi < nrow; ++yi) {
                    const int rb = __builtin_amdgcn_readfirstlane(SB.rowb[yi]);
                    if (rcur < 0) {
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int j = 0; j < 7; ++j)
#pragma unroll
                                for (int k = 0; k < 4; ++k) gwin[d][j][k] = RA_G_AT(rb + d, j, k);
                        rcur = rb;
                    }
                    while (rcur < rb) {  // slide the window down by one bin row
#pragma unroll
                        for (int j = 0; j < 7; ++j)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                gwin[0][j][k] = gwin[1][j][k];
                                gwin[1][j][k] = gwin[2][j][k];
                                gwin[2][j][k] = RA_G_AT(rcur + 3, j, k);
                            }
                        ++rcur;
                    }
                    const float w0 = SB.roww[0][yi], w1 = SB.roww[1][yi], w2 = SB.roww[2][yi];
#pragma unroll
                    for (int j = 0; j < 7; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            SB.tb[j][k][lane] = __builtin_fmaf(w2, gwin[2][j][k], __builtin_fmaf(w1, gwin[1][j][k], w0 * gwin[0][j][k]));
                    // (a lane reads back only what it wrote: no wave synchronisation needed beyond program order)
                    float t0[4], t1[4], t2[4];
                    int bcur = __builtin_amdgcn_readfirstlane(S.colb[0]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        t0[k] = bcur < P ? SB.tb[bcur < P ? bcur : 0][k][lane] : 0.f;
                        t1[k] = bcur + 1 < P ? SB.tb[bcur + 1 < P ? bcur + 1 : 0][k][lane] : 0.f;
                        t2[k] = bcur + 2 < P ? SB.tb[bcur + 2 < P ? bcur + 2 : 0][k][lane] : 0.f;
                    }
                    float* frow = feat + ((size_t)(ys0 + yi) * W + xs0) * C + cb0 + lane;
                    for (int xi = 0; xi < ncol; ++xi) {
                        const int cbx = __builtin_amdgcn_readfirstlane(S.colb[xi]);
                        while (bcur < cbx) {
                            ++bcur;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                t0[k] = t1[k]; t1[k] = t2[k];
                                t2[k] = bcur + 2 < P ? SB.tb[bcur + 2 < P ? bcur + 2 : 0][k][lane] : 0.f;
                            }
                        }
                        const float c0w = S.colw[0][xi], c1w = S.colw[1][xi], c2w = S.colw[2][xi];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float v = __builtin_fmaf(c2w, t2[k], __builtin_fmaf(c1w, t1[k], c0w * t0[k]));
                            if (chok[k] && v != 0.f) atomicAdd(frow + (size_t)xi * C + k * 64, v);
                        }
                    }
                }
            }
#undef RA_G_AT
            return;
        }
    }
    // Channel mapping: lane l takes channels l, l+64, l+128, ... so that one wave-wide atomic instruction covers 64 consecutive
    // floats (four full 64-byte lines) instead of touching 16 lines with four lanes each.
    for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw)
            for (int cb = 0; cb < C; cb += 256) {
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ch = cb + k * 64 + lane;
                    g[k] = ch < C ? osr_to_float(dout[(size_t)(ph * P + pw) * C + ch]) / count : 0.f;
                }
                if (!fallback) {
                    const int y0 = S.lo[0][ph], ny = S.n[0][ph], x0 = S.lo[1][pw], nx = S.n[1][pw];
                    for (int j = 0; j < ny; ++j) {
                        const float wy = S.w[0][ph][j];
                        float* row = feat + ((size_t)(y0 + j) * W + x0) * C + cb + lane;
                        for (int i = 0; i < nx; ++i) {
                            const float wgt = wy * S.w[1][pw][i];
                            if (wgt == 0.f) continue;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (cb + k * 64 + lane < C) atomicAdd(row + (size_t)i * C + k * 64, wgt * g[k]);
                        }
                    }
                } else {
                    for (int iy = 0; iy < gh; ++iy) {
                        int yl, yh; float hy, ly;
                        if (!axis_sample(sh, ph, bh, iy, gh, H, &yl, &yh, &hy, &ly)) continue;
                        for (int ix = 0; ix < gw; ++ix) {
                            int xl, xh; float hx, lx;
                            if (!axis_sample(sw, pw, bw, ix, gw, W, &xl, &xh, &hx, &lx)) continue;
                            const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int ch = cb + k * 64 + lane;
                                if (ch >= C) continue;
                                atomicAdd(feat + ((size_t)yl * W + xl) * C + ch, w1 * g[k]);
                                atomicAdd(feat + ((size_t)yl * W + xh) * C + ch, w2 * g[k]);
                                atomicAdd(feat + ((size_t)yh * W + xl) * C + ch, w3 * g[k]);
                                atomicAdd(feat + ((size_t)yh * W + xh) * C + ch, w4 * g[k]);
                            }
                        }
                    }
                }
            }
}

extern "C" osr_status osr_roi_align_bwd(const osr_pyramid* dfeat, int32_t n, const float* boxes, const int32_t* batch_idx, int64_t m,
                                        int32_t pooled, int32_t canonical_level, int32_t canonical_size, int32_t min_level, const void* dout,
                                        int32_t dout_dtype, void* stream) {
    OSR_REQUIRE(dfeat && boxes && batch_idx && dout, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd: null pointer");
    OSR_REQUIRE(dfeat->num_levels >= 1 && dfeat->num_levels <= 4, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd: 1..4 levels, got %d", dfeat->num_levels);
    OSR_REQUIRE(pooled >= 1 && pooled <= 7, OSR_ERR_UNSUPPORTED, "osr_roi_align_bwd: pooled size 1..7, got %d", pooled);
    OSR_REQUIRE(dfeat->c > 0 && dfeat->c % 4 == 0, OSR_ERR_UNSUPPORTED, "osr_roi_align_bwd: channels must be a multiple of 4, got %d", dfeat->c);
    OSR_REQUIRE(osr_dtype_ok(dout_dtype), OSR_ERR_INVALID_ARG, "osr_roi_align_bwd: bad dtype");
    OSR_REQUIRE(n >= 1 && m >= 0 && m < (1ll << 31) && canonical_size > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd: bad n / m / canonical_size");
    if (m == 0) return OSR_OK;
    RoiAlignBwdArgs a;
    for (int l = 0; l < 4; ++l) {
        int s = l < dfeat->num_levels ? l : 0;
        OSR_REQUIRE(dfeat->data[s] && dfeat->h[s] > 0 && dfeat->w[s] > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd: bad level %d", s);
        a.data[l] = (float*)dfeat->data[s]; a.h[l] = dfeat->h[s]; a.w[l] = dfeat->w[s]; a.scale[l] = dfeat->scale[s];
    }
    a.num_levels = dfeat->num_levels; a.c = dfeat->c; a.boxes = boxes; a.batch_idx = batch_idx; a.m = m;
    a.pooled = pooled; a.canonical_level = canonical_level; a.canonical_size = canonical_size; a.min_level = min_level;
    a.dout = dout;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((m + RA_BWD_WPB - 1) / RA_BWD_WPB)), block(RA_BWD_WPB * 64);
    switch (dout_dtype) {
        case OSR_F32: hipLaunchKernelGGL(roi_align_bwd_kernel<float>, grid, block, 0, st, a); break;
        case OSR_F16: hipLaunchKernelGGL(roi_align_bwd_kernel<f16_t>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(roi_align_bwd_kernel<bf16_t>, grid, block, 0, st, a); break;
    }
    OSR_CHECK_LAUNCH("osr_roi_align_bwd");
    return OSR_OK;
}
