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

// Table sizes set the LDS footprint of a wave and with it the occupancy: 5.3 KB per wave -> the kernel is limited by its
// registers (5 waves per SIMD), not by LDS (4 with the 64 / 352 tables of round 1: 1.51 -> 1.34 ms on the bench's proposals).
#ifndef RA_MAXC
#define RA_MAXC 32  // table columns per bin. Bins of this model's pyramid are at most 7 px wide (28 px RoIs on p2 .. 1333 px on p5);
                    // a wider bin (single-level pyramids in the tests) takes the per-sample loop
#endif
#ifndef RA_MAXX
#define RA_MAXX 192 // (backward) columns of the whole RoI footprint on the streaming path
#endif
#ifndef RA_FWD_MAXX
#define RA_FWD_MAXX 96  // (forward) steps of the streamed (shorter) side of the footprint; beyond: the per-bin loop. This model's pyramid: <= 44
#endif
#ifndef RA_PAIR
#define RA_PAIR 0   // 1: 2-byte features take two steps per wave instruction (16-byte loads, ra_bin_row_pair). Rounds 2-5 ran it: with the
                    // dead prefetches of the old loops the kernel sat at the vector-memory instruction rate and halving the instructions paid.
                    // With exact-length streams (ra_stream, round 6) the instruction count is 2.3x lower and the binding resource is
                    // the L1-miss traffic (5.4 GB per pass from L2 at 0.6-0.7 of the L2 -> L1 ceiling) under memory latency: the
                    // one-step form's 28 accumulator registers (against 56) let four or five waves per SIMD in instead of three, which is worth
                    // more than the wider loads (same box: 1.20 ms round 5, 1.15 pair / 3 waves, 1.11 one-step / 4 waves, 1.08 / 5 waves; RA_MINW below)
#endif
#ifndef RA_PD1
#define RA_PD1 6  // pipeline depth of the pair path by pixels per step (NY = 1..4)
#define RA_PD2 4
#define RA_PD3 2
#define RA_PD4 2
#endif
#ifndef RA_SD1
#define RA_SD1 8  // pipeline depth of the one-step-per-instruction path by pixels per step (NY = 1, 2, 3, 4-5, 6): with the
#define RA_SD2 4  // (2 D - 1) NY loads of 8 bytes per lane a stream holds: 94 registers
#define RA_SD3 3
#define RA_SD4 2
#define RA_SD6 2
#endif
#define RA_MAXD 16  // zero rows after the last step of S.wfull (the pair path's odd last step reads one row past; until round 6 the streams ran up to D steps past the footprint)

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
    const int* order;  // processing order of the RoIs (a permutation of 0..m-1) or null: the result does not depend on it
    const int* order_nvalid;  // (with order) how many leading entries of it are real RoIs, the rest padding rows; null: unknown
    int no_pad_fill;   // 1: padding rows (batch index < 0) are left unwritten instead of zero-filled (osr_roi_align_fwd_ordered_ex)
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
    __attribute__((aligned(16))) float wfull[RA_FWD_MAXX + RA_MAXD][8];  // per step of the streamed axis: its weight in each of the 7 bins (0 outside the
                                                                     // bin's footprint, and in the RA_MAXD rows after the last step)
};

#ifndef RA_WPR
#define RA_WPR 1  // waves per RoI: the bin rows (columns) of the inner axis are dealt over the waves of the workgroup, which share the
                  // RoI's tables. 1 = a wave per RoI (RA_WPB RoIs per workgroup, no workgroup barriers).
#endif
#ifndef RA_WPB
#define RA_WPB 2  // RoIs per workgroup when RA_WPR == 1 (2: -3 % against 4 once the small tables let five waves per SIMD in)
#endif
#if RA_WPR > 1
#undef RA_WPB
#define RA_WPB 1
#define RA_SYNC() __syncthreads()
#define RA_ANY(x) __syncthreads_or(x)
#else
#define RA_SYNC() ra_wave_sync()
#define RA_ANY(x) __any(x)
#endif
#define RA_THREADS (RA_WPB * RA_WPR * 64)
typedef float ra_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int ra_u2 __attribute__((ext_vector_type(2)));

// Register image of one pixel's 4 channels (this lane's share), loaded through a buffer resource: the address is the resource
// base + a per-lane byte offset (constant for the whole RoI) + a wave-uniform byte offset computed on the scalar unit, so the
// loads of the streaming loop cost no vector instructions beyond themselves.
template <class T> struct Buf4;
template <> struct Buf4<float> {
    ra_u32x4 r;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int vo, int so) { r = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0); }
    __device__ __forceinline__ void get(ra_f2& lo, ra_f2& hi) const {
        lo = ra_f2{__uint_as_float(r[0]), __uint_as_float(r[1])}; hi = ra_f2{__uint_as_float(r[2]), __uint_as_float(r[3])};
    }
};
template <> struct Buf4<f16_t> {
    ra_u2 r;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int vo, int so) { r = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0); }
    __device__ __forceinline__ void get(ra_f2& lo, ra_f2& hi) const {
        typedef f16_t h4 __attribute__((ext_vector_type(4)));
        const h4 t = __builtin_bit_cast(h4, r);
        lo = ra_f2{(float)t[0], (float)t[1]}; hi = ra_f2{(float)t[2], (float)t[3]};
    }
};
template <> struct Buf4<bf16_t> {
    ra_u2 r;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int vo, int so) { r = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0); }
    __device__ __forceinline__ void get(ra_f2& lo, ra_f2& hi) const {
        lo = ra_f2{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u)};
        hi = ra_f2{__uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
    }
};

// A software-pipelined stream of n steps with D groups of loads in flight and NO load past the last step (round 6).
// Until then the loops below prefetched D groups ahead with the step index clamped to the last one and ran in whole D-groups, so every
// stream issued D + (rounding) groups of dead loads (L1 hits with zero weights). On the bench's proposals (scripts/sim_roi_loads.py:
// seven streams per RoI of 6-7 wave steps in the median against D = 2..6) those were 56 % of all load instructions -- 445 per RoI
// against 194 -- in a kernel that is bound by the CU's vector-memory instruction rate, and a load that is out of the buffer's range
// costs 3/4 of one that hits L1 (scripts/exp_ta_oob.hip: 16 against 22 cycles per wave instruction), so the dead ones have to go, not
// just miss. n = q D + r: the first q D steps run the classic rotation over the D register groups v (every load unconditional: the
// compiler's counted waits stay exact); the last r < D steps get register groups of their own (u), loaded -- behind wave-uniform
// branches -- in front of the drain of v and consumed after it. (A conditional reload of v inside the drain would make every later
// wait of the drain assume it was not issued, i.e. wait for it: one full memory latency per step.) Steps are consumed in ascending
// order, as before: same summation order, same values.
template <int D, class G, class LoadF, class ConsF>
__device__ __forceinline__ void ra_stream(int n, LoadF load, ConsF cons) {
    G v[D], u[D > 1 ? D - 1 : 1];
    const int q = n / D, r = n - q * D;  // (wave-uniform)
    if (q >= 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) load(v[d], d);
        for (int i = 0; i + 1 < q; ++i) {
#pragma unroll
            for (int d = 0; d < D; ++d) { cons(v[d], i * D + d); load(v[d], (i + 1) * D + d); }
        }
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) if (d < r) load(u[d], q * D + d);
    if (q >= 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) cons(v[d], (q - 1) * D + d);
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) if (d < r) cons(u[d], q * D + d);
}

template <class TO>
__device__ __forceinline__ void ra_store_bins(const ra_f2 (&acc)[7][2], float inv_count, TO* __restrict__ outrow, size_t ostride, bool cok, int P) {
    if (!cok) return;
#pragma unroll
    for (int b = 0; b < 7; ++b)
        if (b < P) {
            const float tot[4] = {acc[b][0][0] * inv_count, acc[b][0][1] * inv_count, acc[b][1][0] * inv_count, acc[b][1][1] * inv_count};
            store4<TO>(outrow + (size_t)b * ostride, tot);
        }
}

// One bin of the inner axis of one RoI for this lane's 4 channels, streamed along the outer axis. Per step: NY pixel loads (the
// pixels of the step inside the bin; NY is wave-uniform and a template parameter), their weighted sum with the bin's NY inner
// weights (scalar registers), and 7 multiply-adds of that sum into the 7 bins of the outer axis with the step's row of S.wfull
// (zeros outside a bin's footprint). No branch, no store and no vector address arithmetic inside the loop, so the loads of the
// next D steps are in flight behind counted waits (ra_stream: nothing is loaded past the last step). The 7 bins are stored when
// the stream ends. Summation order per bin: steps ascending, as the reference's ix loop.
template <int NY, class TI, class TO>
__device__ __forceinline__ void ra_bin_row(__amdgpu_buffer_rsrc_t rs, int voff, int base, int istride_b, int sstride_b, int ncol, const RaWaveLds& S,
                                           const float (&wy)[6], float inv_count, TO* __restrict__ outrow, size_t ostride, bool cok, int P) {
    constexpr int D = NY == 1 ? RA_SD1 : NY == 2 ? RA_SD2 : NY == 3 ? RA_SD3 : NY <= 5 ? RA_SD4 : RA_SD6;  // steps in flight (D * NY loads of 8 or 16 bytes per lane)
    static_assert(D <= RA_MAXD, "S.wfull is padded with RA_MAXD zero rows");
    ra_f2 acc[7][2];
#pragma unroll
    for (int b = 0; b < 7; ++b) { acc[b][0] = ra_f2{0.f, 0.f}; acc[b][1] = ra_f2{0.f, 0.f}; }
    struct Grp { Buf4<TI> p[NY]; };
    ra_stream<D, Grp>(
        ncol,
        [&](Grp& g, int x) {
            const int so = base + x * sstride_b;
#pragma unroll
            for (int j = 0; j < NY; ++j) g.p[j].load(rs, voff, so + j * istride_b);
        },
        [&](Grp& g, int x) {
            ra_f2 c0 = ra_f2{0.f, 0.f}, c1 = ra_f2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                ra_f2 lo, hi;
                g.p[j].get(lo, hi);
                const ra_f2 wj = ra_f2{wy[j], wy[j]};
                c0 = __builtin_elementwise_fma(wj, lo, c0);
                c1 = __builtin_elementwise_fma(wj, hi, c1);
            }
            const float4 wa = *reinterpret_cast<const float4*>(&S.wfull[x][0]), wb = *reinterpret_cast<const float4*>(&S.wfull[x][4]);
            const float wv[7] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z};
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const ra_f2 wq = ra_f2{wv[b], wv[b]};
                acc[b][0] = __builtin_elementwise_fma(wq, c0, acc[b][0]);
                acc[b][1] = __builtin_elementwise_fma(wq, c1, acc[b][1]);
            }
        });
    ra_store_bins<TO>(acc, inv_count, outrow, ostride, cok, P);
}

// 16-byte image of 8 channels of one pixel (2-byte feature types): 32 lanes cover a 256-channel pixel, so one wave instruction
// loads TWO pixels. The vector memory path spends ~16 cycles on a wave instruction of up to 8 bytes per lane and ~21 on one of
// 16 (measured, scripts/exp_ta_width.hip): at 8 bytes per lane the kernel was bound by exactly that (rocprofv3: TA busy 85 %).
template <class T> struct Buf8;
// acc + w * (fp16 half of a packed register), fp32: v_fma_mix_f32 converts inside the multiply-add (the compiler's own choice for
// this pattern is v_cvt_f32_f16 + half a v_pk_fma_f32, 1.5 instructions per element instead of 1)
template <int HI, bool FIRST> __device__ __forceinline__ float ra_mix(unsigned packed, float w, float c) {
    float d;
    if (FIRST) {
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "s"(w));
        else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "s"(w));
    } else {
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "s"(w), "v"(c));
        else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "s"(w), "v"(c));
    }
    return d;
}
template <> struct Buf8<f16_t> {
    ra_u32x4 r;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int vo, int so) { r = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0); }
    template <bool FIRST> __device__ __forceinline__ void fma_into(float w, float (&cs)[8]) const {  // cs += w * pixel (w: wave-uniform)
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {
            cs[2 * i] = ra_mix<0, FIRST>(r[i], w, cs[2 * i]);
            cs[2 * i + 1] = ra_mix<1, FIRST>(r[i], w, cs[2 * i + 1]);
        }
    }
};
template <> struct Buf8<bf16_t> {
    ra_u32x4 r;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int vo, int so) { r = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0); }
    template <bool FIRST> __device__ __forceinline__ void fma_into(float w, float (&cs)[8]) const {
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {
            cs[2 * i] = __builtin_fmaf(__uint_as_float(r[i] << 16), w, FIRST ? 0.f : cs[2 * i]);
            cs[2 * i + 1] = __builtin_fmaf(__uint_as_float(r[i] & 0xffff0000u), w, FIRST ? 0.f : cs[2 * i + 1]);
        }
    }
};
template <> struct Buf8<float> {  // (never instantiated for a load: the pair path is for 2-byte features)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t, int, int) {}
    template <bool FIRST> __device__ __forceinline__ void fma_into(float, float (&cs)[8]) const { _Pragma("unroll") for (int i = 0; i < 8; ++i) cs[i] = 0.f; }
};

// ra_bin_row for 2-byte features, two steps of the outer axis per wave instruction: lanes 0..31 take step 2s, lanes 32..63 step
// 2s + 1 (voff carries the half's extra step), 8 channels per lane. Each half adds its step into its own 7 x 8 accumulators with
// its own row of S.wfull; when the stream ends v_permlane32_swap brings the two halves of a pair of bins together (lanes 0..31
// get the total of the even bin, lanes 32..63 of the odd one) and each half stores its bin: 4 store instructions per bin row.
// The inner weighted sum is written per channel so that it compiles to v_fma_mix_f32 (fp16 operand, fp32 accumulate: one
// instruction per element instead of a convert and half a packed multiply-add). Per bin the summation order is: even steps
// ascending, odd steps ascending, then the two partial sums (fp32; the reference adds all samples in one ascending loop).
template <int NY, class TI, class TO>
__device__ __forceinline__ void ra_bin_row_pair(__amdgpu_buffer_rsrc_t rs, int voff, int base, int istride_b, int sstride_b, int ncol, const RaWaveLds& S,
                                                int half, const float (&wy)[6], float inv_count, TO* __restrict__ outrow, size_t ostride, bool cok, int P) {
    // wave steps in flight (D * NY loads of 16 bytes per lane in the rotation + (D - 1) * NY for the last steps: <= 60 registers)
    constexpr int D = NY == 1 ? RA_PD1 : NY == 2 ? RA_PD2 : NY == 3 ? RA_PD3 : RA_PD4;
    static_assert(2 * D + 1 <= RA_MAXD, "S.wfull is padded with RA_MAXD zero rows");
    ra_f2 acc[7][4];
#pragma unroll
    for (int b = 0; b < 7; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[b][q] = ra_f2{0.f, 0.f};
    struct Grp { Buf8<TI> p[NY]; };
    const int nws = (ncol + 1) >> 1, step2_b = 2 * sstride_b;
    ra_stream<D, Grp>(
        nws,
        [&](Grp& g, int ws) {
            const int so = base + ws * step2_b;
#pragma unroll
            for (int j = 0; j < NY; ++j) g.p[j].load(rs, voff, so + j * istride_b);
        },
        [&](Grp& g, int ws) {
            float cs[8];
            g.p[0].template fma_into<true>(wy[0], cs);
#pragma unroll
            for (int j = 1; j < NY; ++j) g.p[j].template fma_into<false>(wy[j], cs);
            const float* wr = &S.wfull[2 * ws + half][0];
            const float4 wa = *reinterpret_cast<const float4*>(wr), wb = *reinterpret_cast<const float4*>(wr + 4);
            const float wv[7] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z};
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const ra_f2 wq = ra_f2{wv[b], wv[b]};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[b][q] = __builtin_elementwise_fma(wq, ra_f2{cs[2 * q], cs[2 * q + 1]}, acc[b][q]);
            }
        });
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        float tot[8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned x = __float_as_uint(acc[2 * k2][q][e]), y = 2 * k2 + 1 < 7 ? __float_as_uint(acc[(2 * k2 + 1) % 7][q][e]) : 0u;
                const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
                tot[2 * q + e] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * inv_count;
            }
        const int bin = 2 * k2 + half;
        if (cok && bin < P) store8<TO>(outrow + (size_t)bin * ostride, tot);
    }
}

// The same bin row when its footprint is deeper than 6 pixels along the inner axis (long boxes: up to size / 7 + 2 per bin):
// the weighted sum of a step runs over the pixels in chunks of 6 loads in flight, inner weights come from the LDS table.
template <class TI, class TO>
__device__ __forceinline__ void ra_bin_row_tall(__amdgpu_buffer_rsrc_t rs, int voff, int base, int istride_b, int sstride_b, int ncol, const RaWaveLds& S,
                                                const float* wrow, int ny, float inv_count, TO* __restrict__ outrow, size_t ostride, bool cok, int P) {
    ra_f2 acc[7][2];
#pragma unroll
    for (int b = 0; b < 7; ++b) { acc[b][0] = ra_f2{0.f, 0.f}; acc[b][1] = ra_f2{0.f, 0.f}; }
    for (int x = 0; x < ncol; ++x) {
        const int so = base + x * sstride_b;
        ra_f2 c0 = ra_f2{0.f, 0.f}, c1 = ra_f2{0.f, 0.f};
        for (int j0 = 0; j0 < ny; j0 += 6) {
            Buf4<TI> v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j].load(rs, voff, so + min(j0 + j, ny - 1) * istride_b);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                ra_f2 lo, hi;
                v[j].get(lo, hi);
                const float w1 = j0 + j < ny ? wrow[j0 + j] : 0.f;
                const ra_f2 wj = ra_f2{w1, w1};
                c0 = __builtin_elementwise_fma(wj, lo, c0);
                c1 = __builtin_elementwise_fma(wj, hi, c1);
            }
        }
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const float w1 = S.wfull[x][b];
            const ra_f2 wq = ra_f2{w1, w1};
            acc[b][0] = __builtin_elementwise_fma(wq, c0, acc[b][0]);
            acc[b][1] = __builtin_elementwise_fma(wq, c1, acc[b][1]);
        }
    }
    ra_store_bins<TO>(acc, inv_count, outrow, ostride, cok, P);
}

// One wave per RoI (RA_WPB RoIs per workgroup, no workgroup barriers). The wave builds the per-axis weight tables in its
// private LDS slice, picks the shorter side of the footprint as the streamed axis, and then, for each of the 7 bins of the
// other axis, walks the footprint one pixel column (or row) at a time: the pixels of the step that fall into the bin are
// reduced with the bin's weights (software pipelined, the next step's loads in flight) and the sum goes into a 3-bin
// sliding window of register accumulators along the streamed axis.
#ifndef RA_MINW
#define RA_MINW (RA_PAIR ? 3 : 4)  // waves per SIMD the register allocation must allow (pair path: 168 registers). One-step path: four, not the
                                   // five its 94 registers would allow -- end to end (bench.py, four passes in flight, same box, two rounds:
                                   // scripts/ab_bench6.sh) 1600-1606 img/s with four against 1602-1605 with five, 1585-1595 with the pair
                                   // path and 1583-1588 with round 5's kernel; the kernel alone 1.094-1.10 / 1.11-1.12 / 1.14-1.19 / 1.21-1.25 ms;
                                   // five waves keep 5 120 RoIs in flight against 32 MiB of L2: hit rate 0.64 instead of 0.73, +0.6 GB from HBM
#endif
template <class TI, class TO>
__global__ __launch_bounds__(RA_THREADS, (sizeof(TI) == 4 ? 3 : RA_MINW)) void roi_align_kernel(RoiAlignArgs a) {  // (fp32 features: 16-byte pixel images, 3 waves)
    __shared__ RaWaveLds s_all[RA_WPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // table builders of one RoI: the whole workgroup (RA_WPR > 1) or the RoI's wave; sub = this wave's share of the bin rows
    const int grp = RA_WPR > 1 ? 0 : wid, sub = RA_WPR > 1 ? wid : 0, gtid = RA_WPR > 1 ? tid : lane;
    // XCD-aware order: workgroup b runs on XCD b % 8, and each XCD walks one CONTIGUOUS share of the list, so RoIs that are
    // neighbours in the list (a.order: in the image) are neighbours in time on one L2. When the list says where its padding rows
    // start (a.order_nvalid), every XCD gets an eighth of the real RoIs followed by its part of the padding, i.e. the same amount
    // of work: with the padding (a fifth of the bench's list, all at the end of the locality order) in the plain eighths the
    // last XCDs idle.
    long long r;
    {
        const int nwg = gridDim.x, bq = blockIdx.x, q = nwg >> 3, rr = nwg & 7, xcd = bq & 7, idx = bq >> 3;
        const int slot0 = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) * RA_WPB;  // first list slot of this XCD
        const int pos = idx * RA_WPB + grp;                                                       // this RoI's slot inside the XCD's share
        r = (long long)slot0 + pos;
        if (a.order_nvalid) {  // this XCD's slots [slot0, slot0 + cap) take the same fraction of the real RoIs as of all slots
            const long long cap = (long long)(xcd < rr ? q + 1 : q) * RA_WPB, total = (long long)nwg * RA_WPB;
            const long long V = min(max(*a.order_nvalid, 0), (int)a.m);
            const long long v0 = V * slot0 / total, v1 = V * (slot0 + cap) / total;  // v1 - v0 <= cap because V <= total
            const long long nv = v1 - v0, p0 = slot0 - v0;                           // padding rows taken by the XCDs before this one
            r = pos < nv ? v0 + pos : V + p0 + (pos - nv);
        }
    }
    if (r >= a.m) return;
    if (a.order) r = a.order[r];
    RaWaveLds& S = s_all[grp];
    const int P = a.pooled, C = a.c;
    const int CE = C;  // element stride between two bins of a row
#define RA_CBASE(c0) ((size_t)(c0))
    TO* out = reinterpret_cast<TO*>(a.out) + (size_t)r * P * P * C;

    const int b = a.batch_idx[r];
    if (b < 0) {  // padding row: zeros (or nothing: the caller never reads it)
        if (a.no_pad_fill) return;
        for (int i = gtid * 4; i < P * P * C; i += RA_WPR * 64 * 4) {
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
    for (int e = gtid; e < 2 * 7 * tcols; e += RA_WPR * 64) {
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
    const bool fallback = RA_ANY(overflow);
#if RA_WPR == 1
    ra_wave_sync();
#endif

    // ---- streaming fast path. The footprint is walked along one axis (the "outer" axis, one step per pixel column or row);
    //      per step the pixels of the other ("inner") axis that fall into the current bin are reduced with the bin's weights,
    //      and the result is added into the 7 bins of the outer axis with the step's weights (S.wfull: zero outside a bin's
    //      footprint, so nothing is assumed about how the bins overlap). The outer axis is the SHORTER side of the footprint:
    //      a step has a fixed cost however few pixels it reduces, and the proposals are 2-10x wider than tall (or the reverse)
    //      often enough that streaming the long side doubled the kernel's time. ----
    int lo_l[2] = {0x7fffffff, 0x7fffffff}, hi_l[2] = {0, 0};
    if (lane < P) {
#pragma unroll
        for (int ax = 0; ax < 2; ++ax) {
            const int lo = S.lo[ax][lane], n = S.n[ax][lane];
            if (n > 0) { lo_l[ax] = lo; hi_l[ax] = lo + n; }
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
        ax_ok[ax] = !fallback && ext_n[ax] <= RA_FWD_MAXX;
    }
    const unsigned long long lvl_bytes = (unsigned long long)H * W * C * sizeof(TI);  // one image of this level: the buffer resource's range
    // outer (streamed) axis: the shorter side when it fits the step table, else the other one
#ifndef RA_AXIS_SELECT
#define RA_AXIS_SELECT 1
#endif
    int oa = (!RA_AXIS_SELECT || ext_n[1] <= ext_n[0]) ? 1 : 0;
    if (!ax_ok[oa]) oa ^= 1;

    if (ax_ok[oa] && lvl_bytes < (1ull << 31)) {
        const int ia = oa ^ 1;
        const int os = ext_lo[oa], nstep = ext_n[oa];
        for (int sl = gtid; sl < nstep + RA_MAXD; sl += RA_WPR * 64) {  // per-step table: the weight of this outer pixel in each bin
            const int x = os + sl;
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                float wv = 0.f;
                if (bb < P && sl < nstep) { const int i = x - S.lo[oa][bb]; if (i >= 0 && i < S.n[oa][bb]) wv = S.w[oa][bb][i]; }
                S.wfull[sl][bb] = wv;
            }
        }
        RA_SYNC();
        const float inv_count = 1.0f / count;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(feat), 0, (int)lvl_bytes, 0x00020000);
        // byte strides of one step along the inner / outer axis, and the element stride between two consecutive output bins of the outer axis
        const int rowstride_b = W * C * (int)sizeof(TI), pix_b = C * (int)sizeof(TI);
        const int istride_b = ia == 0 ? rowstride_b : pix_b, sstride_b = ia == 0 ? pix_b : rowstride_b;
        const size_t ostride = oa == 1 ? (size_t)CE : (size_t)P * CE;
        for (int pb = sub; pb < P; pb += RA_WPR) {  // bins along the inner axis
            const int i0 = __builtin_amdgcn_readfirstlane(S.lo[ia][pb]), ni = __builtin_amdgcn_readfirstlane(S.n[ia][pb]);
            float wi[6];  // wave-uniform inner weights (live in scalar registers)
#pragma unroll
            for (int j = 0; j < 6; ++j) wi[j] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(j < ni ? S.w[ia][pb][j] : 0.f)));
            const int base = i0 * istride_b + os * sstride_b;
            if (RA_PAIR && sizeof(TI) == 2 && C % 8 == 0 && ni >= 1 && ni <= 4 && nstep > 0) {  // two steps per wave instruction, 8 channels per lane
                const int half = lane >> 5;
                for (int cb0 = 0; cb0 < C; cb0 += 256) {
                    const int c0 = cb0 + (lane & 31) * 8;
                    const bool cok = c0 < C;
                    const int voff = (cok ? c0 : 0) * (int)sizeof(TI) + half * sstride_b;
                    TO* outrow = out + (size_t)pb * (oa == 1 ? (size_t)P * CE : (size_t)CE) + RA_CBASE(c0);
                    switch (ni) {
                        case 1: ra_bin_row_pair<1, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, half, wi, inv_count, outrow, ostride, cok, P); break;
                        case 2: ra_bin_row_pair<2, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, half, wi, inv_count, outrow, ostride, cok, P); break;
                        case 3: ra_bin_row_pair<3, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, half, wi, inv_count, outrow, ostride, cok, P); break;
                        default: ra_bin_row_pair<4, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, half, wi, inv_count, outrow, ostride, cok, P); break;
                    }
                }
                continue;
            }
            for (int cb0 = 0; cb0 < C; cb0 += 256) {
                const int c0 = cb0 + lane * 4;
                const bool cok = c0 < C;
                const int voff = (cok ? c0 : 0) * (int)sizeof(TI);
                TO* outrow = out + (size_t)pb * (oa == 1 ? (size_t)P * CE : (size_t)CE) + RA_CBASE(c0);
                switch (nstep > 0 ? (ni > 6 ? 7 : ni) : 0) {  // (no step: the pipelined loop would have nothing valid to prefetch)
                    case 1: ra_bin_row<1, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 2: ra_bin_row<2, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 3: ra_bin_row<3, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 4: ra_bin_row<4, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 5: ra_bin_row<5, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 6: ra_bin_row<6, TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, wi, inv_count, outrow, ostride, cok, P); break;
                    case 7: ra_bin_row_tall<TI, TO>(rs, voff, base, istride_b, sstride_b, nstep, S, S.w[ia][pb], ni, inv_count, outrow, ostride, cok, P); break;
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
    for (int ph = sub; ph < P; ph += RA_WPR) {
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
                store4<TO>(out + (size_t)(ph * P + pw) * CE + RA_CBASE(c0), acc);
            }
        }
    }
}

template <class TI>
static osr_status launch_out(const RoiAlignArgs& a, int out_dtype, hipStream_t st) {
    dim3 grid((unsigned)((a.m + RA_WPB - 1) / RA_WPB)), block(RA_THREADS);
    switch (out_dtype) {
        case OSR_F32: hipLaunchKernelGGL((roi_align_kernel<TI, float>), grid, block, 0, st, a); break;
        case OSR_F16: hipLaunchKernelGGL((roi_align_kernel<TI, f16_t>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((roi_align_kernel<TI, bf16_t>), grid, block, 0, st, a); break;
    }
    OSR_CHECK_LAUNCH("osr_roi_align_fwd");
    return OSR_OK;
}

static osr_status roi_align_fwd_impl(const osr_pyramid* f, int32_t feat_dtype, int32_t n, const float* boxes,
                                     const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                     int32_t canonical_size, int32_t min_level, const int32_t* order,
                                     const int32_t* order_nvalid, void* out, int32_t out_dtype, void* stream, int32_t flags = 0) {
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
    a.out = out; a.order = order; a.order_nvalid = order ? order_nvalid : nullptr;
    a.no_pad_fill = (flags & OSR_ROI_NO_PADDING_FILL) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    switch (feat_dtype) {
        case OSR_F32: return launch_out<float>(a, out_dtype, st);
        case OSR_F16: return launch_out<f16_t>(a, out_dtype, st);
        default: return launch_out<bf16_t>(a, out_dtype, st);
    }
}

extern "C" osr_status osr_roi_align_fwd_ordered(const osr_pyramid* f, int32_t feat_dtype, int32_t n, const float* boxes,
                                                const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                                int32_t canonical_size, int32_t min_level, const int32_t* order,
                                                const int32_t* order_nvalid, void* out, int32_t out_dtype, void* stream) {
    return roi_align_fwd_impl(f, feat_dtype, n, boxes, batch_idx, m, pooled, canonical_level, canonical_size, min_level, order, order_nvalid,
                              out, out_dtype, stream);
}

extern "C" osr_status osr_roi_align_fwd_ordered_ex(const osr_pyramid* f, int32_t feat_dtype, int32_t n, const float* boxes,
                                                   const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                                   int32_t canonical_size, int32_t min_level, const int32_t* order,
                                                   const int32_t* order_nvalid, int32_t flags, void* out, int32_t out_dtype, void* stream) {
    OSR_REQUIRE((flags & ~OSR_ROI_NO_PADDING_FILL) == 0, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_ordered_ex: unknown flag bits 0x%x", flags);
    return roi_align_fwd_impl(f, feat_dtype, n, boxes, batch_idx, m, pooled, canonical_level, canonical_size, min_level, order, order_nvalid,
                              out, out_dtype, stream, flags);
}

extern "C" osr_status osr_roi_align_fwd(const osr_pyramid* f, int32_t feat_dtype, int32_t n, const float* boxes,
                                        const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                        int32_t canonical_size, int32_t min_level, void* out, int32_t out_dtype,
                                        void* stream) {
    return osr_roi_align_fwd_ordered(f, feat_dtype, n, boxes, batch_idx, m, pooled, canonical_level, canonical_size, min_level, nullptr, nullptr,
                                     out, out_dtype, stream);
}

// ------------------------------------------------------------------------------------------------------
// Locality order of the RoI list. The forward kernel's HBM traffic is dominated by re-reads: the proposals of an image overlap
// each other ~8x (rocprofv3 FETCH_SIZE: 5.7 GB per 16-image pass against 0.73 GB of pyramid), and in score order the RoIs that
// are resident on an XCD at one time are spread over whole images, far more than its 4 MiB L2 holds. Bucket sort by
// (image, level, 32x32-pixel tile of the box centre on that level): three small launches (count, scan, scatter). Only the
// processing order changes: every RoI still writes its own output row, so results are bit-identical for any order.
// ------------------------------------------------------------------------------------------------------
#define RA_BUCKETS 128  // per image: 77 + 24 + 6 + 2 tiles for an 800x1333 pyramid; ids beyond are clamped (a hint, not a contract)

struct RoiOrderArgs {
    int h[4], w[4];
    float scale[4];
    int num_levels, n;
    const float* boxes;
    const int* batch_idx;
    long long m;
    int canonical_level, canonical_size, min_level;
    int* bucket;   // (m) scratch
    int* counts;   // (n * RA_BUCKETS + 2): [0 .. nb) buckets, nb = padding rows
    int* order;
};

__device__ __forceinline__ int ra_bucket_of(const RoiOrderArgs& a, long long r) {
    const int b = a.batch_idx[r];
    if (b < 0 || b >= a.n) return a.n * RA_BUCKETS;
    const float4 bx = *reinterpret_cast<const float4*>(a.boxes + r * 4);
    float sz = sqrtf((bx.z - bx.x) * (bx.w - bx.y));
    float lvf = floorf((float)a.canonical_level + log2f(sz / (float)a.canonical_size + 1e-8f));
    lvf = fminf(fmaxf(lvf, (float)a.min_level), (float)(a.min_level + a.num_levels - 1));
    const int lv = (int)lvf - a.min_level;
    int base = 0;
    for (int l = 0; l < lv; ++l) base += ((a.h[l] + 31) >> 5) * ((a.w[l] + 31) >> 5);
    const int tnx = (a.w[lv] + 31) >> 5, tny = (a.h[lv] + 31) >> 5;
    const float cx = 0.5f * (bx.x + bx.z) * a.scale[lv], cy = 0.5f * (bx.y + bx.w) * a.scale[lv];
    const int tx = min(max((int)cx >> 5, 0), tnx - 1), ty = min(max((int)cy >> 5, 0), tny - 1);  // NaN -> 0
    return b * RA_BUCKETS + min(base + ty * tnx + tx, RA_BUCKETS - 1);
}

// (Padding rows all share one bucket: a wave adds its padding rows with ONE atomic, or the 10^4 same-address atomics of a padded
// list serialise into ~0.15 ms.)
__global__ __launch_bounds__(256) void roi_order_count(RoiOrderArgs a) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k = r < a.m ? ra_bucket_of(a, r) : -1, kpad = a.n * RA_BUCKETS;
    if (r < a.m) a.bucket[r] = k;
    const unsigned long long pm = __ballot(k == kpad);
    if (k == kpad) {
        if ((int)(threadIdx.x & 63) == __ffsll((long long)pm) - 1) atomicAdd(&a.counts[kpad], __popcll(pm));
    } else if (k >= 0) atomicAdd(&a.counts[k], 1);
}

__global__ __launch_bounds__(1024) void roi_order_scan(int* counts, int nb, int* nvalid) {  // exclusive scan in place, one workgroup; the
                                                                                            // padding bucket (the last) starts at *nvalid
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (nb + 1023) / 1024, lo = tid * per, hi = min(lo + per, nb);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; ++i) {
        const int c = counts[i];
        counts[i] = run;
        if (i == nb - 1) *nvalid = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void roi_order_scatter(RoiOrderArgs a) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, kpad = a.n * RA_BUCKETS;
    const int k = r < a.m ? a.bucket[r] : -1;
    const unsigned long long pm = __ballot(k == kpad);
    int pos = 0;
    if (pm) {  // wave-uniform
        const int leader = __ffsll((long long)pm) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&a.counts[kpad], __popcll(pm));
        base = __shfl(base, leader, 64);
        pos = base + __popcll(pm & ((1ull << lane) - 1ull));
    }
    if (k >= 0 && k != kpad) pos = atomicAdd(&a.counts[k], 1);
    if (k >= 0) a.order[pos] = (int)r;
}

extern "C" int64_t osr_roi_locality_order_workspace_bytes(int32_t n, int64_t m) {
    if (n < 1 || m < 0) return 0;
    return ((int64_t)m + (int64_t)n * RA_BUCKETS + 2) * 4;
}

extern "C" osr_status osr_roi_locality_order(const osr_pyramid* f, int32_t n, const float* boxes, const int32_t* batch_idx, int64_t m,
                                             int32_t canonical_level, int32_t canonical_size, int32_t min_level, int32_t* order,
                                             int32_t* nvalid, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(f && boxes && batch_idx && order && nvalid && workspace, OSR_ERR_INVALID_ARG, "osr_roi_locality_order: null pointer");
    OSR_REQUIRE(f->num_levels >= 1 && f->num_levels <= 4, OSR_ERR_INVALID_ARG, "osr_roi_locality_order: 1..4 levels, got %d", f->num_levels);
    OSR_REQUIRE(n >= 1 && m >= 0 && m < (1ll << 31) && canonical_size > 0, OSR_ERR_INVALID_ARG, "osr_roi_locality_order: bad n/m/canonical_size");
    OSR_REQUIRE((((uintptr_t)boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_roi_locality_order: boxes must be 16-byte aligned");
    OSR_REQUIRE(workspace_bytes >= osr_roi_locality_order_workspace_bytes(n, m), OSR_ERR_WORKSPACE, "osr_roi_locality_order: workspace needs %lld bytes",
                (long long)osr_roi_locality_order_workspace_bytes(n, m));
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        OSR_REQUIRE(hipMemsetAsync(nvalid, 0, 4, st) == hipSuccess, OSR_ERR_LAUNCH, "osr_roi_locality_order: memset failed");
        return OSR_OK;
    }
    RoiOrderArgs a;
    for (int l = 0; l < 4; ++l) {
        const int s = l < f->num_levels ? l : 0;
        OSR_REQUIRE(f->h[s] > 0 && f->w[s] > 0, OSR_ERR_INVALID_ARG, "osr_roi_locality_order: bad level %d", s);
        a.h[l] = f->h[s]; a.w[l] = f->w[s]; a.scale[l] = f->scale[s];
    }
    a.num_levels = f->num_levels; a.n = n; a.boxes = boxes; a.batch_idx = batch_idx; a.m = m;
    a.canonical_level = canonical_level; a.canonical_size = canonical_size; a.min_level = min_level;
    a.counts = (int*)workspace; a.bucket = a.counts + (size_t)n * RA_BUCKETS + 2; a.order = order;
    const int nb = n * RA_BUCKETS + 1;
    OSR_REQUIRE(hipMemsetAsync(a.counts, 0, (size_t)(nb + 1) * 4, st) == hipSuccess, OSR_ERR_LAUNCH, "osr_roi_locality_order: memset failed");
    const unsigned grid = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(roi_order_count, dim3(grid), dim3(256), 0, st, a);
    OSR_CHECK_LAUNCH("osr_roi_locality_order(count)");
    hipLaunchKernelGGL(roi_order_scan, dim3(1), dim3(1024), 0, st, a.counts, nb, nvalid);
    OSR_CHECK_LAUNCH("osr_roi_locality_order(scan)");
    hipLaunchKernelGGL(roi_order_scatter, dim3(grid), dim3(256), 0, st, a);
    OSR_CHECK_LAUNCH("osr_roi_locality_order(scatter)");
    return OSR_OK;
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
struct RaBwdTables {
    float w[2][7][RA_MAXC];  // [axis: 0 = y, 1 = x][bin][column of the bin's footprint]
    int lo[2][8], n[2][8];
    int colb[RA_MAXX];       // per footprint column: first unfinished bin
    float colw[3][RA_MAXX];  // weight of the column in bins colb, colb+1, colb+2
};
struct RaBwdLds {
    RaBwdTables t;               // per-axis bin tables + per-column (x) window table
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
    RaBwdTables& S = SB.t;
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

// ------------------------------------------------------------------------------------------------------
// RoIAlign backward, pixel-centric ("dense") form: no atomics, no zero-initialised output, bitwise reproducible.
// The scatter form above pays one fp32 atomic per footprint pixel and channel (2.4 GB of atomics per training step of 16 x 512 RoIs at
// the 1.3 TB/s the memory side adds at, plus the 1.5 GB zero fill of the gradient pyramid). Here one workgroup owns an 8 x 8 pixel
// tile of one level of one image for all channels (wave w: channels 64 w .. 64 w + 63, lane = channel, the tile's 64 sums in
// registers) and GATHERS: it tests the image's RoIs (two per thread: level + a conservative footprint box against the tile),
// compacts the hits in list order, and for every hit evaluates
//     d feat[y][x] += sum_by wy[y][by] * (sum_bx wx[x][bx] * g[by][bx])        (g = d out / sample count)
// with the per-axis weights wy / wx of the tile's 8 rows / columns in each of the 7 bins (112 threads compute them from the same
// axis_sample() the forward uses; a sample outside [-1, size] has weight 0 on its axis, hence in the product, as in the reference).
// Bins without weight in the tile are skipped wave-uniformly. Every tile is written exactly once (zeros where no RoI reaches).
// The RoI list must be image-major with a fixed stride (rows [b S, (b + 1) S) belong to image b; batch_idx < 0 = padding).
// ------------------------------------------------------------------------------------------------------
#define RD_T 8
struct RoiBwdDenseArgs {
    void* data[4];
    int h[4], w[4];
    float scale[4];
    int tiles_x[4], tiles_y[4], tile_off[5];  // tile grid of one image per level; first workgroup of each level
    int num_levels, c, n, S;
    const float* boxes;
    const int* batch_idx;
    int pooled, canonical_level, canonical_size, min_level;
    const void* dout;
};

template <class TG, class TO>
#ifndef RD_MINW
#define RD_MINW 4    // waves per SIMD the register allocation must allow (128 registers, no scratch)
#endif
__global__ __launch_bounds__(256, RD_MINW) void roi_align_bwd_dense_kernel(RoiBwdDenseArgs a) {
    __shared__ float s_w[2][2][RD_T][8];  // [buffer][axis: 0 = y, 1 = x][pixel row / column of the tile][bin] (axis 0 already / count)
    __shared__ unsigned short s_hits[1024];
    __shared__ unsigned char s_flag[1024];
    __shared__ int s_nhit;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (Plain workgroup order on purpose. An XCD-aware order -- every XCD walking one contiguous run of the tile list, as the conv and
    // weight-gradient kernels do -- measured 0.73 -> 0.87 ms on uniform boxes and 0.80 -> 1.02 ms on the sampler's clustered positives:
    // the tiles under a ground-truth box carry most of the work, and a contiguous run puts them all on one XCD; round-robin spreads them.)
    const int bid = (int)blockIdx.x;
    int lv = 0;
    while (lv + 1 < a.num_levels && bid >= a.tile_off[lv + 1]) ++lv;
    const int tpi = a.tiles_x[lv] * a.tiles_y[lv];
    const int rel = bid - a.tile_off[lv];
    const int b = rel / tpi, tt = rel - b * tpi;
    const int ty0 = (tt / a.tiles_x[lv]) * RD_T, tx0 = (tt % a.tiles_x[lv]) * RD_T;
    const int H = a.h[lv], W = a.w[lv], P = a.pooled, C = a.c;
    const float scale = a.scale[lv];
    const float lmin = (float)a.min_level, lmax = (float)(a.min_level + a.num_levels - 1);

    // ---- 1. which of the image's RoIs reach this tile (list order is kept: the summation order is fixed) ----
    for (int j = tid; j < a.S; j += 256) {
        const long long r = (long long)b * a.S + j;
        bool hit = false;
        if (a.batch_idx[r] == b) {
            const float bx1 = a.boxes[r * 4 + 0], by1 = a.boxes[r * 4 + 1], bx2 = a.boxes[r * 4 + 2], by2 = a.boxes[r * 4 + 3];
            const float sz = sqrtf((bx2 - bx1) * (by2 - by1));
            float lvf = floorf((float)a.canonical_level + log2f(sz / (float)a.canonical_size + 1e-8f));
            lvf = fminf(fmaxf(lvf, lmin), lmax);
            if ((int)lvf - a.min_level == lv) {
                const float sw = bx1 * scale - 0.5f, sh = by1 * scale - 0.5f, ew = bx2 * scale - 0.5f, eh = by2 * scale - 0.5f;
                // samples lie in (start, end); a sample touches floor(v) and floor(v) + 1, and v in [-1, 0] is pulled to 0
                const float ylo = floorf(fminf(sh, eh)) - 1.f, yhi = floorf(fmaxf(sh, eh)) + 2.f;
                const float xlo = floorf(fminf(sw, ew)) - 1.f, xhi = floorf(fmaxf(sw, ew)) + 2.f;
                hit = yhi >= (float)ty0 && ylo <= (float)(ty0 + RD_T - 1) && xhi >= (float)tx0 && xlo <= (float)(tx0 + RD_T - 1);
            }
        }
        s_flag[j] = hit ? 1 : 0;
    }
    __syncthreads();
    if (wid == 0) {
        int base = 0;
        for (int j0 = 0; j0 < a.S; j0 += 64) {
            const int j = j0 + lane;
            const bool f = j < a.S && s_flag[j];
            const unsigned long long m = __ballot(f);
            if (f) s_hits[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)j;
            base += __popcll(m);
        }
        if (lane == 0) s_nhit = base;
    }
    __syncthreads();
    const int nhit = s_nhit;

    float acc[RD_T * RD_T];
#pragma unroll
    for (int i = 0; i < RD_T * RD_T; ++i) acc[i] = 0.f;
    const int ch = wid * 64 + lane;
    const bool chok = ch < C;
    for (int k = 0; k < nhit; ++k) {
        const int buf = k & 1;
        const long long r = (long long)b * a.S + s_hits[k];
        const float bx1 = a.boxes[r * 4 + 0], by1 = a.boxes[r * 4 + 1], bx2 = a.boxes[r * 4 + 2], by2 = a.boxes[r * 4 + 3];
        const float sw = bx1 * scale - 0.5f, sh = by1 * scale - 0.5f, ew = bx2 * scale - 0.5f, eh = by2 * scale - 0.5f;
        const float rw = ew - sw, rh = eh - sh;
        const float bw = rw / (float)P, bh = rh / (float)P;
        const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
        if (tid < 2 * RD_T * 7) {
            const int axis = tid / (RD_T * 7), rem = tid - axis * RD_T * 7, pix = rem / 7, bin = rem - pix * 7;
            float wsum = 0.f;
            if (bin < P) {
                const float start = axis ? sw : sh, bs = axis ? bw : bh;
                const int grid = axis ? gw : gh, size = axis ? W : H, pp = (axis ? tx0 : ty0) + pix;
                for (int i = 0; i < grid; ++i) {
                    int lo, hi; float wl, wh;
                    if (!axis_sample(start, bin, bs, i, grid, size, &lo, &hi, &wl, &wh)) continue;
                    if (lo == pp) wsum += wl;
                    if (hi == pp) wsum += wh;
                }
                if (axis == 0) wsum /= (float)max(gh * gw, 1);
            }
            s_w[buf][axis][pix][bin] = wsum;
        }
        __syncthreads();  // (the buffer written two hits ago was read before the previous hit's barrier)
        // Which bins have weight inside this tile (the same in every lane: scalar). Lanes 0..55 look at entry (pixel lane / 7, bin lane % 7) of
        // each axis; bit p * 7 + b of the ballot = "pixel p has weight in bin b".
        unsigned long long my = 0ull, mx = 0ull;
        {
            const int pp = lane / 7, bb = lane - pp * 7;
            const bool in = lane < RD_T * 7;
            my = __ballot(in && s_w[buf][0][pp][bb] != 0.f);
            mx = __ballot(in && s_w[buf][1][pp][bb] != 0.f);
        }
        unsigned acty = 0u, actx = 0u;
#pragma unroll
        for (int bb = 0; bb < 7; ++bb) {
            const unsigned long long col = 0x0002040810204081ull << bb;  // bits bb, 7 + bb, .., 49 + bb: the 8 pixels of bin bb
            acty |= (my & col) ? (1u << bb) : 0u;
            actx |= (mx & col) ? (1u << bb) : 0u;
        }
        if (acty == 0u || actx == 0u) continue;  // (wave-uniform; the hit test is conservative: no weight in this tile after all)
        // Round 6: the bins with weight form a contiguous range of rows rb0..rb1 (and columns). The old loop loaded one value of g and used
        // it at once -- one exposed L2 latency per (by, bx) pair, up to 49 per hit, and a tile under a ground-truth box is hit by a
        // hundred sampled RoIs. Now a row of g (all 7 bx, unconditionally: a bin without weight multiplies a zero) is in flight while
        // the previous row is multiplied in; the loads are unconditional so that the compiler's counted waits leave the next row in flight.
        const int rb0 = __builtin_ctz(acty), rb1 = 31 - __builtin_clz(acty);
        const TG* g = reinterpret_cast<const TG*>(a.dout) + (size_t)r * P * P * C + (chok ? ch : 0);
        float cur[7], nxt[7];
#define RD_LOAD_ROW(dst, byy)                                                                                       \
        _Pragma("unroll") for (int bx = 0; bx < 7; ++bx) dst[bx] = osr_to_float(g[(size_t)((byy) * P + min(bx, P - 1)) * C]);
        RD_LOAD_ROW(cur, rb0)
        for (int by = rb0; by <= rb1; ++by) {
            const int byn = min(by + 1, rb1);
            // (compiler fence: the next row's loads go out HERE, ahead of this row's arithmetic, and the x weights are re-read from LDS per row
            // instead of being hoisted into 56 registers across the row loop -- hoisted: 166 registers, three waves per SIMD and 1.33 ms;
            // with the fence 128 registers, four waves and 0.73 ms; the loop before round 6: 1.16 ms. scripts/ab_roibwd6.sh, same box)
            asm volatile("" ::: "memory");
            RD_LOAD_ROW(nxt, byn)
            float trow[RD_T];
#pragma unroll
            for (int x = 0; x < RD_T; ++x) {
                const float4 w0 = *reinterpret_cast<const float4*>(&s_w[buf][1][x][0]), w1 = *reinterpret_cast<const float4*>(&s_w[buf][1][x][4]);
                float t = w0.x * cur[0];
                t = __builtin_fmaf(w0.y, cur[1], t); t = __builtin_fmaf(w0.z, cur[2], t); t = __builtin_fmaf(w0.w, cur[3], t);
                t = __builtin_fmaf(w1.x, cur[4], t); t = __builtin_fmaf(w1.y, cur[5], t); t = __builtin_fmaf(w1.z, cur[6], t);
                trow[x] = t;
            }
#pragma unroll
            for (int y = 0; y < RD_T; ++y) {
                const float wyv = s_w[buf][0][y][by];
#pragma unroll
                for (int x = 0; x < RD_T; ++x) acc[y * RD_T + x] = __builtin_fmaf(wyv, trow[x], acc[y * RD_T + x]);
            }
#pragma unroll
            for (int bx = 0; bx < 7; ++bx) cur[bx] = nxt[bx];
        }
#undef RD_LOAD_ROW
    }
    // ---- 3. the tile, once ----
    if (chok) {
        TO* out = reinterpret_cast<TO*>(a.data[lv]) + (size_t)b * H * W * C + ch;
#pragma unroll
        for (int y = 0; y < RD_T; ++y)
#pragma unroll
            for (int x = 0; x < RD_T; ++x)
                if (ty0 + y < H && tx0 + x < W) out[((size_t)(ty0 + y) * W + tx0 + x) * C] = osr_from_float<TO>(acc[y * RD_T + x]);
    }
}

extern "C" osr_status osr_roi_align_bwd_dense(const osr_pyramid* dfeat, int32_t n, const float* boxes, const int32_t* batch_idx, int64_t m,
                                              int32_t rois_per_image, int32_t pooled, int32_t canonical_level, int32_t canonical_size,
                                              int32_t min_level, const void* dout, int32_t dout_dtype, int32_t out_dtype, void* stream) {
    OSR_REQUIRE(dfeat && boxes && batch_idx && dout, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd_dense: null pointer");
    OSR_REQUIRE(out_dtype == OSR_F32 || out_dtype == dout_dtype, OSR_ERR_UNSUPPORTED, "osr_roi_align_bwd_dense: out_dtype must be f32 or dout's dtype");
    OSR_REQUIRE(dfeat->num_levels >= 1 && dfeat->num_levels <= 4, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd_dense: 1..4 levels, got %d", dfeat->num_levels);
    OSR_REQUIRE(pooled >= 1 && pooled <= 7, OSR_ERR_UNSUPPORTED, "osr_roi_align_bwd_dense: pooled size 1..7, got %d", pooled);
    OSR_REQUIRE(dfeat->c > 0 && dfeat->c <= 256, OSR_ERR_UNSUPPORTED, "osr_roi_align_bwd_dense: at most 256 channels, got %d", dfeat->c);
    OSR_REQUIRE(osr_dtype_ok(dout_dtype), OSR_ERR_INVALID_ARG, "osr_roi_align_bwd_dense: bad dtype");
    OSR_REQUIRE(n >= 1 && m >= 0 && canonical_size > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd_dense: bad n / m / canonical_size");
    OSR_REQUIRE(rois_per_image >= 0 && rois_per_image <= 1024 && m == (int64_t)n * rois_per_image, OSR_ERR_UNSUPPORTED,
                "osr_roi_align_bwd_dense: the RoI list must be image-major with a fixed stride <= 1024 (m = %lld, n = %d, stride %d)", (long long)m, n, rois_per_image);
    RoiBwdDenseArgs a;
    long long off = 0;
    for (int l = 0; l < 4; ++l) {
        const int s = l < dfeat->num_levels ? l : 0;
        OSR_REQUIRE(dfeat->data[s] && dfeat->h[s] > 0 && dfeat->w[s] > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_bwd_dense: bad level %d", s);
        a.data[l] = (void*)dfeat->data[s]; a.h[l] = dfeat->h[s]; a.w[l] = dfeat->w[s]; a.scale[l] = dfeat->scale[s];
        a.tiles_x[l] = (a.w[l] + RD_T - 1) / RD_T; a.tiles_y[l] = (a.h[l] + RD_T - 1) / RD_T;
        a.tile_off[l] = (int)off;
        if (l < dfeat->num_levels) off += (long long)n * a.tiles_x[l] * a.tiles_y[l];
    }
    a.tile_off[4] = (int)off;
    for (int l = dfeat->num_levels; l < 4; ++l) a.tile_off[l] = (int)off;
    OSR_REQUIRE(off < (1ll << 31), OSR_ERR_UNSUPPORTED, "osr_roi_align_bwd_dense: too many tiles");
    a.num_levels = dfeat->num_levels; a.c = dfeat->c; a.n = n; a.S = rois_per_image; a.boxes = boxes; a.batch_idx = batch_idx;
    a.pooled = pooled; a.canonical_level = canonical_level; a.canonical_size = canonical_size; a.min_level = min_level;
    a.dout = dout;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)off), block(256);
    const bool lowp_out = out_dtype != OSR_F32;
    switch (dout_dtype) {
        case OSR_F32: hipLaunchKernelGGL((roi_align_bwd_dense_kernel<float, float>), grid, block, 0, st, a); break;
        case OSR_F16:
            if (lowp_out) hipLaunchKernelGGL((roi_align_bwd_dense_kernel<f16_t, f16_t>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((roi_align_bwd_dense_kernel<f16_t, float>), grid, block, 0, st, a);
            break;
        default:
            if (lowp_out) hipLaunchKernelGGL((roi_align_bwd_dense_kernel<bf16_t, bf16_t>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((roi_align_bwd_dense_kernel<bf16_t, float>), grid, block, 0, st, a);
            break;
    }
    OSR_CHECK_LAUNCH("osr_roi_align_bwd_dense");
    return OSR_OK;
}
