// RoIAlign over the FPN pyramid, tile-centric form for gfx950 (include/osr.h: osr_roi_align_fwd_tiled).
//
// Replaces the same reference call as osr_roi_align.hip -- [d2] ROIPooler.forward + torchvision roi_align(aligned=True,
// sampling_ratio=0) at /root/reference/openset_rcnn/modeling/roi_heads/osrcnn_roi_heads.py:108-113,306 -- for the proposal lists of
// inference, where the RoIs of an image overlap each other 7-50x (measured on the benchmark's proposals: 16.0 M footprint pixels
// over 1.43 M pyramid pixels). The wave-per-RoI kernel re-reads every overlap through the CU's vector memory pipe (10.9 GB per
// 16-image step at 35 GB/s per CU: the pace of its L1 miss queue); here a workgroup stages a REGION of one pyramid level once in LDS
// and pools every RoI whose footprint lies inside it from there.
//
// Design (MI355X):
//  * Channel-sliced regions. 160 KB of LDS hold 50 x 96 pixels of 16 channels (32 B per pixel). So that a 16-channel slice of a
//    pixel run is contiguous in HBM (a 32-byte piece of a 512-byte NHWC pixel would drag a whole 128-byte line through L2 -> L1),
//    the kernel reads a PLANAR copy of the pyramid, (n, c/16, h, w, 16), which the FPN output convolutions write beside the NHWC
//    tensor (osr_conv_params.out2). A task = (region, slice); regions sit on a half-overlapping grid (stride 25 x 48), a RoI belongs
//    to the region that holds its footprint's top-left pixel, and the (few) RoIs whose footprint does not fit a region, is wider than
//    64 columns or has bins taller than 7 pixels are left to the wave-per-RoI kernel (same output buffer).
//  * Staging by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction), one barrier, no registers.
//  * The x contraction on the matrix cores. sum over the samples of a bin of bilinear taps = sum_r wy[ph][r] sum_c wx[pw][c] f[r][c]
//    (per-axis weight tables, exact w.r.t. torchvision's validity / clamp rules, as in osr_roi_align.hip). For one footprint row the
//    inner sum over columns is D[ch][pw] = F^T W: v_mfma_f32_16x16x32_f16 with A = the row's pixels (k = 32 columns, m = 16
//    channels; read from the [pixel][channel] LDS image with the transposing ds_read_b64_tr_b16) and B = the RoI's x weights, fp32
//    weights split into fp16 hi (columns n = 0..6) and lo (n = 8..14) parts so that the products are exact to 2^-22. The y contraction
//    is 4 fp32 FMAs per MFMA with a wave-uniform weight. The hi and lo halves are added once per RoI (DPP row rotate).
//  * Output rows are SLICE-MAJOR: out[roi][slice][bin][16 channels], so a task's result for a RoI is 1568 contiguous bytes (fp16).
//    The consumer (FC1) takes its K axis in that order (host/weights.pack_fc1_weight(slice_major=16)).
//  * A plan pass (thread per RoI: level, footprint, region; counting sort of the RoIs by region) and a descriptor pass (wave per
//    RoI: the B fragments, the y table, bin row ranges) run in front; descriptors live in the caller's workspace.
// Numerics: fp32 accumulation; differs from the reference loop in summation order and in the 2^-22 relative rounding of the x
// weights (tolerance 1e-4 of the row maximum, asserted in tests/test_roi_tiled.py). A non-finite feature value poisons the bins of
// every RoI row whose 32-column window holds it (0 * inf), not only the bins that sample it; finite features: no difference.
#include "osr_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void rt_lds_void_t;
typedef short rt_s16x4 __attribute__((ext_vector_type(4)));
typedef short rt_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) rt_s16x4 rt_lds_s16x4_t;
typedef f16_t rt_h8 __attribute__((ext_vector_type(8)));
typedef f16_t rt_h4 __attribute__((ext_vector_type(4)));
typedef float rt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rt_u32x4 __attribute__((ext_vector_type(4)));

#define RT_CS 16                      // channels per slice (one MFMA M tile)
#define RT_RH 50                      // region rows
#define RT_RW 96                      // region columns (multiple of 32: three 1-KiB DMA pieces per row)
#define RT_ROWB (RT_RW * RT_CS * 2)   // bytes of a region row in LDS
#define RT_PADROWS 3                  // zeroed rows behind the staged ones: a bin's rows go through the MFMA four at a time
#define RT_LDS_CTR ((RT_RH + RT_PADROWS) * RT_ROWB)  // the workgroup's RoI counter
#define RT_LDS (RT_LDS_CTR + 64)      // 162 880 bytes
#define RT_WY 9                       // rows of a bin's y table that can be non-zero
#define RT_DESC_WORDS 704             // descriptor stride in 4-byte words: [0,256) B fragment of K step 0, [256,512) K step 1,
#define RT_DESC_WY 512                //   [512,640) y table (bin * 16 + row; zero from the bin's last row on),
#define RT_DESC_HDR 640               //   [640] k0, [641] nK, [642,649) first row of each bin (region-relative), [649,656) its rows
#define RT_OOB 0x80000000u
#define RT_WAVES 16

struct RtLevel {
    const void* planes;  // (n, c/16, h, w, 16) fp16
    int h, w;
    float scale;
    int gh, gw, sh, sw;  // region grid of one image: gh x gw regions, origin (i * sh, j * sw) clamped into the map
    int base;            // id of the level's first region
};

struct RtArgs {
    RtLevel lv[4];
    int num_levels, n, c, nsl;
    const float* boxes;
    const int* batch_idx;
    long long m;
    int canonical_level, canonical_size, min_level;
    int nr, nrp;      // regions, and that count rounded up to a multiple of 8
    int* rid;         // (m) region of a RoI on this path; -1: left to the wave-per-RoI kernel; -2: padding row
    int* counts;      // (nr)
    int* offsets;     // (nr)
    int* cursor;      // (nr)
    int* tag;         // (m) plan -> scatter: region | (K steps - 1) << 30 of the RoIs on this path
    int* list;        // (m) RoI ids grouped by region, (K steps - 1) in bit 30
    unsigned* desc;   // (m * RT_DESC_WORDS)
    void* out;
};

// One sample coordinate of torchvision's pre_calc_for_bilinear_interpolate along one axis (the same arithmetic as axis_sample of
// osr_roi_align.hip). Returns false when the sample is outside [-1, size] (contributes nothing).
__device__ __forceinline__ bool rt_axis_sample(float start, int bin, float bin_size, int i, int grid, int size, int* lo, int* hi, float* wl, float* wh) {
    float v = start + bin * bin_size + ((float)i + .5f) * bin_size / (float)grid;
    if (v < -1.0f || v > (float)size) return false;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= size - 1) { h = l = size - 1; v = (float)l; } else h = l + 1;
    const float f = v - (float)l;
    *lo = l; *hi = h; *wh = f; *wl = 1.f - f;
    return true;
}

struct RtGeom {
    int lv, b;
    float sw, sh, bw, bh;  // start and bin size per axis, in pixels of the level
    int gw, gh;            // samples per bin and axis
    int r_lo, r_hi, c_lo, c_hi;  // a superset of the footprint (inclusive)
    int org_r, org_c, rows, cols, region;  // the region that holds (r_lo, c_lo)
    int k0, nk;
    bool fit;
};

// Geometry of RoI r: level ([d2] assign_boxes_to_levels in fp32, as osr_roi_align.hip), sample grid, a conservative footprint,
// the region it belongs to and whether the tiled path takes it.
__device__ __forceinline__ void rt_geom(const RtArgs& a, long long r, int b, RtGeom& g) {
    const float4 bx = *reinterpret_cast<const float4*>(a.boxes + r * 4);
    const float sz = sqrtf((bx.z - bx.x) * (bx.w - bx.y));
    float lvf = floorf((float)a.canonical_level + log2f(sz / (float)a.canonical_size + 1e-8f));
    lvf = fminf(fmaxf(lvf, (float)a.min_level), (float)(a.min_level + a.num_levels - 1));  // NaN -> min level via fmaxf
    g.lv = (int)lvf - a.min_level;
    g.b = b;
    const RtLevel& L = a.lv[g.lv];
    const int H = L.h, W = L.w;
    g.sw = bx.x * L.scale - 0.5f; g.sh = bx.y * L.scale - 0.5f;
    const float ew = bx.z * L.scale - 0.5f, eh = bx.w * L.scale - 0.5f;
    const float rw = ew - g.sw, rh = eh - g.sh;
    g.bw = rw / 7.f; g.bh = rh / 7.f;
    g.gh = (int)ceilf(rh / 7.f); g.gw = (int)ceilf(rw / 7.f);
    g.fit = false;
    g.region = -1;
    if (!(g.gh >= 1 && g.gw >= 1 && g.gh <= 64 && g.gw <= 64)) return;  // empty / degenerate / non-finite boxes: the other kernel
    if (!(g.bh <= 7.0f)) return;  // a bin then spans at most RT_WY rows: floor(bh * (1 - 1/gh)) + 3 <= 9
    // first and last sample of each axis (the same expression as rt_axis_sample): the samples are monotone in between
    const float vx0 = g.sw + .5f * g.bw / (float)g.gw, vx1 = g.sw + 6 * g.bw + ((float)(g.gw - 1) + .5f) * g.bw / (float)g.gw;
    const float vy0 = g.sh + .5f * g.bh / (float)g.gh, vy1 = g.sh + 6 * g.bh + ((float)(g.gh - 1) + .5f) * g.bh / (float)g.gh;
    if (!(vx1 >= -1.0f && vx0 <= (float)W && vy1 >= -1.0f && vy0 <= (float)H)) return;  // no valid sample (or NaN)
    g.c_lo = min((int)fminf(fmaxf(vx0, 0.f), (float)W), W - 1);
    g.c_hi = min((int)fminf(fmaxf(vx1, 0.f), (float)W) + 1, W - 1);
    g.r_lo = min((int)fminf(fmaxf(vy0, 0.f), (float)H), H - 1);
    g.r_hi = min((int)fminf(fmaxf(vy1, 0.f), (float)H) + 1, H - 1);
    const int fw = g.c_hi - g.c_lo + 1;
    if (fw > 64) return;
    const int i = min(g.r_lo / L.sh, L.gh - 1), j = min(g.c_lo / L.sw, L.gw - 1);
    g.org_r = min(i * L.sh, max(H - RT_RH, 0));
    g.org_c = min(j * L.sw, max(W - RT_RW, 0));
    g.rows = min(RT_RH, H); g.cols = min(RT_RW, W);
    if (g.r_hi >= g.org_r + g.rows || g.c_hi >= g.org_c + g.cols) return;
    g.nk = fw <= 32 ? 1 : 2;
    if (g.cols < 32 * g.nk) return;
    g.k0 = min(g.c_lo - g.org_c, g.cols - 32 * g.nk);  // the K window [k0, k0 + 32 nk) lies inside the staged columns and holds the footprint
    g.region = L.base + (b * L.gh + i) * L.gw + j;
    g.fit = true;
}

// counters[id] += 1 for the active lanes; returns the value before this lane's add. One pass over the wave's most common case: the
// lanes that share the first active lane's id add once (a wave of the list is 64 consecutive proposals of one image and level; on
// the coarse levels, where a region is the whole map, they all share one counter, and 10^4 adds to one 64-byte line take 0.1 ms),
// the others add for themselves in one vector instruction.
__device__ __forceinline__ int rt_counter_add(int* __restrict__ counters, int id, bool active) {
    const int lane = threadIdx.x & 63;
    const unsigned long long act = __ballot(active);
    if (!act) return 0;
    const int leader = __ffsll((long long)act) - 1;
    const int lid = __shfl(id, leader, 64);
    const bool grp = active && id == lid;
    const unsigned long long same = __ballot(grp);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counters[lid], __popcll(same));
    base = __shfl(base, leader, 64);
    if (grp) return base + __popcll(same & ((1ull << lane) - 1ull));
    if (active) return atomicAdd(&counters[id], 1);
    return 0;
}

// ---- plan: region of every RoI, RoIs per region ----
__global__ __launch_bounds__(256) void rt_plan_kernel(RtArgs a) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    int id = -3;
    if (r < a.m) {
        const int b = a.batch_idx[r];
        id = -1;
        if (b < 0) id = -2;
        else if (b < a.n) {
            RtGeom g;
            rt_geom(a, r, b, g);
            if (g.fit) id = g.region | ((g.nk - 1) << 30);
        }
        a.rid[r] = id < 0 ? id : (id & 0x3fffffff);
    }
    rt_counter_add(a.counts, id & 0x3fffffff, id >= 0);
    if (id >= 0) a.tag[r] = id;
}

// ---- scatter: the list of every region; an entry is the RoI id with its number of K steps - 1 in bit 30 ----
__global__ __launch_bounds__(256) void rt_scatter_kernel(RtArgs a) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool on = r < a.m && a.rid[r] >= 0;
    const int t = on ? a.tag[r] : 0, id = t & 0x3fffffff;
    const int pos = rt_counter_add(a.cursor, id, on);
    if (on) a.list[a.offsets[id] + pos] = (int)r | (t & 0x40000000);
}

// exclusive scan of counts[0, nr) into offsets, one workgroup
__global__ __launch_bounds__(1024) void rt_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int nr) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (nr + 1023) / 1024, lo = tid * per, hi = min(lo + per, nr);
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
    for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
}

// ---- descriptors: one wave per RoI of the tiled path ----
__global__ __launch_bounds__(256) void rt_desc_kernel(RtArgs a) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.m) return;
    const int id = a.rid[r];
    if (id < 0) return;
    RtGeom g;
    rt_geom(a, r, a.batch_idx[r], g);  // (deterministic: the same answer as in the plan pass)
    const RtLevel& L = a.lv[g.lv];
    unsigned* __restrict__ d = a.desc + (size_t)r * RT_DESC_WORDS;
    // B fragments: lane (n = lane & 15, kg = lane >> 4) holds W[bin = n & 7][column k0 + 32 st + 8 kg + j], j = 0..7; n < 8: the fp16
    // hi part, n >= 8: the lo part (w - hi); bin 7 is a zero column. W = plain sum of the samples' taps (the 1 / count goes into wy).
    const int n = lane & 15, kg = lane >> 4, bin = n & 7, part = n >> 3;
    for (int st = 0; st < g.nk; ++st) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = 0.f;
        const int col0 = g.org_c + g.k0 + 32 * st + 8 * kg;
        if (bin < 7)
            for (int i = 0; i < g.gw; ++i) {
                int lo, hi; float wl, wh;
                if (!rt_axis_sample(g.sw, bin, g.bw, i, g.gw, L.w, &lo, &hi, &wl, &wh)) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    w[j] += lo == col0 + j ? wl : 0.f;
                    w[j] += hi == col0 + j ? wh : 0.f;
                }
            }
        rt_h8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f16_t h = (f16_t)w[j];
            f[j] = part ? (f16_t)(w[j] - (float)h) : h;
        }
        *reinterpret_cast<rt_u32x4*>(d + st * 256 + lane * 4) = __builtin_bit_cast(rt_u32x4, f);
    }
    // y side: row range of every bin (lanes 0..6), then the table (lane e -> bin e / 9, row e % 9 of the bin), scaled by 1 / count
    int first = -1, last = -1;
    if (lane < 7)
        for (int i = 0; i < g.gh; ++i) {
            int lo, hi; float wl, wh;
            if (!rt_axis_sample(g.sh, lane, g.bh, i, g.gh, L.h, &lo, &hi, &wl, &wh)) continue;
            if (first < 0) first = lo;
            last = hi;
        }
    const int rs_l = first < 0 ? g.org_r : first, rc_l = first < 0 ? 0 : min(last - first + 1, RT_WY);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = lane + 64 * half, e_bin = min(e >> 4, 6), e_t = e & 15;
        const int rs_e = __shfl(rs_l, e_bin, 64), rc_e = __shfl(rc_l, e_bin, 64);
        float wy = 0.f;
        if ((e >> 4) < 7 && e_t < rc_e) {
            const int row = rs_e + e_t;
            for (int i = 0; i < g.gh; ++i) {
                int lo, hi; float wl, wh;
                if (!rt_axis_sample(g.sh, e_bin, g.bh, i, g.gh, L.h, &lo, &hi, &wl, &wh)) continue;
                wy += lo == row ? wl : 0.f;
                wy += hi == row ? wh : 0.f;
            }
            wy *= 1.0f / (float)max(g.gh * g.gw, 1);
        }
        d[RT_DESC_WY + e] = __float_as_uint(wy);
    }
    if (lane < 7) { d[RT_DESC_HDR + 2 + lane] = (unsigned)(rs_l - g.org_r); d[RT_DESC_HDR + 9 + lane] = (unsigned)rc_l; }
    if (lane == 7) d[RT_DESC_HDR] = (unsigned)g.k0;
    if (lane == 8) d[RT_DESC_HDR + 1] = (unsigned)g.nk;
}

__device__ __forceinline__ float rt_ror8(float v) {  // lane n of each row of 16 <- lane (n + 8) % 16
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
}

template <class TO> __device__ __forceinline__ void rt_store4(TO* p, const float v[4]);
template <> __device__ __forceinline__ void rt_store4<float>(float* p, const float v[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void rt_store4<f16_t>(f16_t* p, const float v[4]) {
    const rt_h4 t = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
    *reinterpret_cast<rt_h4*>(p) = t;
}

__device__ __forceinline__ float rt_add_ror8(float v) {  // v + (lane (n + 8) % 16 of the same row of 16)'s v, one instruction
    float r;
    asm("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

// The x and y contractions of one RoI for the staged slice: acc[ph] = sum over the bin's rows of wy * (F_row^T W). A bin's rows go
// through the matrix cores four at a time (independent MFMAs: the chain LDS read -> MFMA -> FMA is latency, not throughput); the
// table's entries behind the bin's last row are zero and the rows behind the region's last one are zeroed LDS.
template <bool TWO>
__device__ __forceinline__ void rt_pool(const unsigned char* lds, int colA, int colB, rt_h8 b0, rt_h8 b1, int hdr_l, int wy_a, int wy_b,
                                        float (&acc)[7][4]) {
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) {
        const int rs = __builtin_amdgcn_readlane(hdr_l, 2 + ph), rc = __builtin_amdgcn_readlane(hdr_l, 9 + ph);
        const unsigned char* pa = lds + rs * RT_ROWB + colA;
        const unsigned char* pb = lds + rs * RT_ROWB + colB;
        const int wsel = ph < 4 ? wy_a : wy_b;
        for (int t0 = 0; t0 < rc; t0 += 4) {
            rt_s16x4 x0[4], x1[4], y0[4], y1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rt_lds_s16x4_t*)(pa + u * RT_ROWB));
                x1[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rt_lds_s16x4_t*)(pb + u * RT_ROWB));
                if (TWO) {  // second K step: columns k0 + 32 .. k0 + 63, 32 positions = 1 KiB further
                    y0[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rt_lds_s16x4_t*)(pa + u * RT_ROWB + 32 * RT_CS * 2));
                    y1[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rt_lds_s16x4_t*)(pb + u * RT_ROWB + 32 * RT_CS * 2));
                }
            }
            rt_f32x4 dd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const rt_s16x8 xa = {x0[u][0], x0[u][1], x0[u][2], x0[u][3], x1[u][0], x1[u][1], x1[u][2], x1[u][3]};
                dd[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rt_h8, xa), b0, rt_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (TWO) {
                    const rt_s16x8 ya = {y0[u][0], y0[u][1], y0[u][2], y0[u][3], y1[u][0], y1[u][1], y1[u][2], y1[u][3]};
                    dd[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rt_h8, ya), b1, dd[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wy = __int_as_float(__builtin_amdgcn_readlane(wsel, (ph & 3) * 16 + t0 + u));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ph][e] = fmaf(wy, dd[u][e], acc[ph][e]);
            }
            pa += 4 * RT_ROWB; pb += 4 * RT_ROWB;
        }
    }
}

// ---- main: one workgroup per (slice, region) ----
template <class TO>
__global__ __launch_bounds__(RT_WAVES * 64) void rt_main_kernel(RtArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (slice, region): slice-major; inside a slice every XCD (block % 8) walks one contiguous range of region ids, so the
    // half-overlapping neighbours of a region are staged on the same L2 at about the same time
    const int s = blockIdx.x / a.nrp, q = blockIdx.x % a.nrp;
    const int rg = (q & 7) * (a.nrp >> 3) + (q >> 3);
    if (rg >= a.nr) return;
    const int cnt = a.counts[rg];
    if (cnt == 0) return;
    int l = 0;
    for (int i = 1; i < a.num_levels; ++i) l = rg >= a.lv[i].base ? i : l;
    const RtLevel& L = a.lv[l];
    const int H = L.h, W = L.w;
    const int local = rg - L.base, j = local % L.gw, t2 = local / L.gw, i = t2 % L.gh, img = t2 / L.gh;
    const int org_r = min(i * L.sh, max(H - RT_RH, 0)), org_c = min(j * L.sw, max(W - RT_RW, 0));
    const int rows = min(RT_RH, H), cols = min(RT_RW, W);

    // ---- stage the region's slice: [row][position][16 channels]; a pixel of column c sits at position c ^ ((c >> 1) & 4) (odd
    //      8-pixel groups rotated by 4: the transposed fragment reads below are then bank-conflict free for every window start) ----
    {
        const unsigned plane_bytes = (unsigned)H * W * (RT_CS * 2);
        const char* plane = reinterpret_cast<const char*>(L.planes) + ((size_t)img * a.nsl + s) * plane_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(plane), 0, (int)plane_bytes, 0x00020000);
        const int npieces = rows * (RT_RW / 32);
        const int pos_l = lane >> 1;
#ifdef RT_DIAG_NO_STAGE  // diagnostic builds only (scripts/ab_roi_tiled.sh): what the kernel costs without its staging / its RoI loop / its stores
        if (false)
#endif
        for (int pi = wid; pi < npieces; pi += RT_WAVES) {
            const int row = pi / (RT_RW / 32), seg = pi - row * (RT_RW / 32);
            const int pos = seg * 32 + pos_l, c = pos ^ ((pos >> 1) & 4);
            const unsigned off = c < cols ? (unsigned)(((org_r + row) * W + org_c + c) * (RT_CS * 2) + (lane & 1) * 16) : RT_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rt_lds_void_t*)(lds + pi * 1024), 16, off, 0, 0, 0);
        }
        if (tid < RT_PADROWS * RT_ROWB / 16) *reinterpret_cast<rt_u32x4*>(lds + rows * RT_ROWB + tid * 16) = rt_u32x4{0u, 0u, 0u, 0u};
        if (tid == 0) *reinterpret_cast<int*>(lds + RT_LDS_CTR) = 2 * RT_WAVES;  // the first two RoIs of every wave are dealt statically
    }

    // ---- the RoIs of the region. Each wave takes the next unclaimed RoI from the workgroup's counter (their costs differ 10x), two
    //      ahead: the list entry of RoI i + 2 and the descriptor of RoI i + 1 are requested before RoI i is computed, the first
    //      descriptor before the staging wait ----
    const int* __restrict__ list = a.list + a.offsets[rg];
    // transposed fragment read (ds_read_b64_tr_b16): lane 4 q4 + pp of a 16-lane group supplies the address of block row q4 (a pixel),
    // columns 4 pp .. 4 pp + 3 (channels); lane i of the group receives channel i of the four pixels. Group kg reads the pixels
    // 8 kg .. 8 kg + 3 (first read) and 8 kg + 4 .. 8 kg + 7 (second) of the 32-column window: the A fragment of v_mfma_f32_16x16x32.
    const int kg = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int n = lane & 15, chg = lane >> 4;
#define RT_LOAD_DESC(E_, HDR_, WYA_, WYB_, B0_, B1_)                                                           \
    {                                                                                                          \
        const unsigned* __restrict__ d_ = a.desc + (size_t)((E_) & 0x3fffffff) * RT_DESC_WORDS;                 \
        HDR_ = (int)d_[RT_DESC_HDR + (lane & 15)];                                                              \
        WYA_ = (int)d_[RT_DESC_WY + lane];                                                                      \
        WYB_ = (int)d_[RT_DESC_WY + 64 + lane];                                                                 \
        B0_ = *reinterpret_cast<const rt_u32x4*>(d_ + lane * 4);                                                \
        B1_ = B0_;                                                                                              \
        if ((E_) & 0x40000000) B1_ = *reinterpret_cast<const rt_u32x4*>(d_ + 256 + lane * 4);                   \
    }
    // (entries: -1 = no more RoIs for this wave)
    int e_cur_v = wid < cnt ? list[wid] : -1, e_nxt_v = wid + RT_WAVES < cnt ? list[wid + RT_WAVES] : -1;
    int hdr_l = 0, wy_a = 0, wy_b = 0;
    rt_u32x4 b0u = {0u, 0u, 0u, 0u}, b1u = {0u, 0u, 0u, 0u};
    int e_cur = __builtin_amdgcn_readfirstlane(e_cur_v);
    if (e_cur >= 0) RT_LOAD_DESC(e_cur, hdr_l, wy_a, wy_b, b0u, b1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef RT_DIAG_NO_COMPUTE
    if (true) return;
#endif
    while (e_cur >= 0) {
        // claim the RoI after next and request its list entry; request the next RoI's descriptor
        int e_nn_v = -1;
        const int e_nxt = __builtin_amdgcn_readfirstlane(e_nxt_v);
        if (e_nxt >= 0) {
            int idx = 0;
            if (lane == 0) idx = __hip_atomic_fetch_add(reinterpret_cast<int*>(lds + RT_LDS_CTR), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx < cnt) e_nn_v = list[idx];
        }
        int hdr_n = 0, wya_n = 0, wyb_n = 0;
        rt_u32x4 b0n = {0u, 0u, 0u, 0u}, b1n = {0u, 0u, 0u, 0u};
        if (e_nxt >= 0) RT_LOAD_DESC(e_nxt, hdr_n, wya_n, wyb_n, b0n, b1n);

        const int r = e_cur & 0x3fffffff;
        const rt_h8 b0 = __builtin_bit_cast(rt_h8, b0u), b1 = __builtin_bit_cast(rt_h8, b1u);
        const int k0 = __builtin_amdgcn_readlane(hdr_l, 0);
        const int ca = k0 + 8 * kg + q4, cb = ca + 4;
        const int colA = ((ca ^ ((ca >> 1) & 4)) * RT_CS + 4 * pp) * 2, colB = ((cb ^ ((cb >> 1) & 4)) * RT_CS + 4 * pp) * 2;
        float acc[7][4];
#pragma unroll
        for (int ph = 0; ph < 7; ++ph)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ph][e] = 0.f;
        if (e_cur & 0x40000000) rt_pool<true>(lds, colA, colB, b0, b1, hdr_l, wy_a, wy_b, acc);
        else rt_pool<false>(lds, colA, colB, b0, b1, hdr_l, wy_a, wy_b, acc);
        // ---- hi + lo (lanes n and n + 8 of a row of 16), store: lane (n < 7, chg) holds channels 4 chg .. 4 chg + 3 of bin (ph, n) ----
        TO* __restrict__ out = reinterpret_cast<TO*>(a.out) + ((size_t)r * a.nsl + s) * (49 * RT_CS) + (n & 7) * RT_CS + 4 * chg;
#pragma unroll
        for (int ph = 0; ph < 7; ++ph) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rt_add_ror8(acc[ph][e]);
#ifdef RT_DIAG_NO_STORE
            if (n < 7 && v[0] == 123.456f)
#else
            if (n < 7)
#endif
                rt_store4<TO>(out + ph * 7 * RT_CS, v);
        }
        e_cur = e_nxt; e_nxt_v = e_nn_v;
        hdr_l = hdr_n; wy_a = wya_n; wy_b = wyb_n; b0u = b0n; b1u = b1n;
    }
#undef RT_LOAD_DESC
}

static osr_dev_mask rt_attr_done;

extern "C" int64_t osr_roi_align_tiled_workspace_bytes(const osr_pyramid* f, int32_t n, int64_t m) {
    if (!f || n < 1 || m < 0 || f->num_levels < 1 || f->num_levels > 4) return 0;
    int64_t nr = 0;
    for (int l = 0; l < f->num_levels; ++l) {
        const int gh = f->h[l] <= RT_RH ? 1 : (f->h[l] - RT_RH + RT_RH / 2 - 1) / (RT_RH / 2) + 1;
        const int gw = f->w[l] <= RT_RW ? 1 : (f->w[l] - RT_RW + RT_RW / 2 - 1) / (RT_RW / 2) + 1;
        nr += (int64_t)n * gh * gw;
    }
    // rid, tag, list (m each) + counts / offsets / cursor (3 nr), 16-byte aligned, + descriptors
    const int64_t ints = ((3 * m + 3 * nr + 3) / 4) * 4;
    return ints * 4 + m * (int64_t)RT_DESC_WORDS * 4;
}

// planes: per level the (n, c/16, h, w, 16) fp16 copy of feats' level. rid_out (m, int32): the region of every RoI the tiled path
// pooled (>= 0), -1 for the RoIs left to osr_roi_align_fwd_masked, -2 for padding rows.
extern "C" osr_status osr_roi_align_fwd_tiled(const osr_pyramid* f, const void* const* planes, int32_t n, const float* boxes,
                                              const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                              int32_t canonical_size, int32_t min_level, void* out, int32_t out_dtype, int32_t** rid_out,
                                              void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(f && planes && boxes && batch_idx && out && workspace, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_tiled: null pointer");
    OSR_REQUIRE(f->num_levels >= 1 && f->num_levels <= 4, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_tiled: 1..4 levels, got %d", f->num_levels);
    OSR_REQUIRE(pooled == 7, OSR_ERR_UNSUPPORTED, "osr_roi_align_fwd_tiled: pooled size 7 only, got %d", pooled);
    OSR_REQUIRE(f->c > 0 && f->c % RT_CS == 0, OSR_ERR_UNSUPPORTED, "osr_roi_align_fwd_tiled: channels must be a multiple of 16, got %d", f->c);
    OSR_REQUIRE(out_dtype == OSR_F16 || out_dtype == OSR_F32, OSR_ERR_UNSUPPORTED, "osr_roi_align_fwd_tiled: out dtype f16 or f32");
    OSR_REQUIRE(n >= 1 && m >= 0 && m < (1ll << 31), OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_tiled: bad n/m");
    OSR_REQUIRE(canonical_size > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_tiled: canonical_size must be > 0");
    OSR_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_tiled: boxes / workspace must be 16-byte aligned");
    OSR_REQUIRE(workspace_bytes >= osr_roi_align_tiled_workspace_bytes(f, n, m), OSR_ERR_WORKSPACE, "osr_roi_align_fwd_tiled: workspace needs %lld bytes",
                (long long)osr_roi_align_tiled_workspace_bytes(f, n, m));
    if (m == 0) return OSR_OK;
    RtArgs a;
    int nr = 0;
    for (int l = 0; l < 4; ++l) {
        const int s = l < f->num_levels ? l : 0;
        OSR_REQUIRE(planes[s] && f->h[s] > 0 && f->w[s] > 0, OSR_ERR_INVALID_ARG, "osr_roi_align_fwd_tiled: bad level %d", s);
        OSR_REQUIRE((long long)f->h[s] * f->w[s] * RT_CS * 2 < (1ll << 31), OSR_ERR_UNSUPPORTED, "osr_roi_align_fwd_tiled: level %d too large", s);
        RtLevel& L = a.lv[l];
        L.planes = planes[s]; L.h = f->h[s]; L.w = f->w[s]; L.scale = f->scale[s];
        L.sh = RT_RH / 2; L.sw = RT_RW / 2;
        L.gh = L.h <= RT_RH ? 1 : (L.h - RT_RH + L.sh - 1) / L.sh + 1;
        L.gw = L.w <= RT_RW ? 1 : (L.w - RT_RW + L.sw - 1) / L.sw + 1;
        L.base = nr;
        if (l < f->num_levels) nr += n * L.gh * L.gw;
    }
    a.num_levels = f->num_levels; a.n = n; a.c = f->c; a.nsl = f->c / RT_CS;
    a.boxes = boxes; a.batch_idx = batch_idx; a.m = m;
    a.canonical_level = canonical_level; a.canonical_size = canonical_size; a.min_level = min_level;
    a.nr = nr; a.nrp = (nr + 7) & ~7;
    int* w = reinterpret_cast<int*>(workspace);
    a.counts = w; a.offsets = w + nr; a.cursor = w + 2 * nr; a.rid = w + 3 * nr; a.tag = a.rid + m; a.list = a.tag + m;
    const int64_t ints = ((3 * m + 3 * (int64_t)nr + 3) / 4) * 4;
    a.desc = reinterpret_cast<unsigned*>(w + ints);
    a.out = out;
    if (rid_out) *rid_out = a.rid;
    hipStream_t st = (hipStream_t)stream;
    OSR_REQUIRE(hipMemsetAsync(a.counts, 0, (size_t)nr * 3 * 4, st) == hipSuccess, OSR_ERR_LAUNCH, "osr_roi_align_fwd_tiled: memset failed");
    hipLaunchKernelGGL(rt_plan_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
    OSR_CHECK_LAUNCH("osr_roi_align_fwd_tiled(plan)");
    hipLaunchKernelGGL(rt_scan_kernel, dim3(1), dim3(1024), 0, st, a.counts, a.offsets, nr);
    OSR_CHECK_LAUNCH("osr_roi_align_fwd_tiled(scan)");
    hipLaunchKernelGGL(rt_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
    OSR_CHECK_LAUNCH("osr_roi_align_fwd_tiled(scatter)");
    hipLaunchKernelGGL(rt_desc_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, a);
    OSR_CHECK_LAUNCH("osr_roi_align_fwd_tiled(descriptors)");
    osr_once_per_device(rt_attr_done, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_main_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_main_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS);
    });
    const dim3 grid((unsigned)a.nrp * a.nsl), block(RT_WAVES * 64);
    if (out_dtype == OSR_F16) hipLaunchKernelGGL((rt_main_kernel<f16_t>), grid, block, RT_LDS, st, a);
    else hipLaunchKernelGGL((rt_main_kernel<float>), grid, block, RT_LDS, st, a);
    OSR_CHECK_LAUNCH("osr_roi_align_fwd_tiled(main)");
    return OSR_OK;
}
