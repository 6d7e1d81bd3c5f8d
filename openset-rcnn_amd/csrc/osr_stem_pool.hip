// ResNet stem in ONE launch for gfx950 (include/osr.h: osr_stem_maxpool_fwd): 7x7 / stride 2 / pad 3 convolution (3 -> 64, FrozenBN
// folded) + ReLU + 3x3 / stride 2 / pad 1 max pool.
//
// Replaces [d2] BasicStem.forward (conv1 -> relu_ -> F.max_pool2d(kernel_size=3, stride=2, padding=1)) of build_resnet_fpn_backbone,
// selected by /root/reference/configs/Base-RCNN-FPN.yaml:3-8 -- until round 4 two launches: the stem as a (kh = 8, kw = 1, cin = 32)
// "view" convolution on the generic implicit-GEMM kernel (K padded 147 -> 256, 0.29 ms) and osr_maxpool3x3s2 (0.16 ms), with the
// 550 MB stem output written to HBM and read back in between.
//
// Design (MI355X): persistent workgroups (4 waves, 51 KB of LDS: three per CU, each walks ~22 tiles at the bench's size; the weights stay
// in registers), one tile at a time = a 4 x 16 tile of POOLED pixels = a 9 x 33 region of stem pixels (the pool's halo is recomputed:
// 297 / 256 = 1.16x) = a 23 x 72-pixel patch of the pre-padded NHWC4 image, staged in LDS once (every input pixel feeds up to 16 stem pixels).
//  * conv: v_mfma_f32_16x16x32 with the WEIGHTS as the A operand (m = 16 output channels; a wave owns the 32 channels of half w & 1 and
//    keeps its 2 x 7 fragments -- one per kernel row -- in registers) and the PIXELS as the B operand (n = 16 stem pixels of the
//    flattened 9 x 33 region -- a wave takes every second group of 16: w >> 1 --, k = one kernel row: 8 pixels x 4 channels = 64
//    contiguous bytes of the patch, one ds_read_b128 per lane; the 8th pixel and the 4th channel meet zero weights). Seven K steps per
//    16 x 16 tile instead of the eight of the padded view, no tile tails (19 groups of 16 for 297 pixels). The group loop is unrolled
//    and a lane's per-group LDS offsets are computed once per workgroup (one packed register per group).
//  * D = W X^T hands a lane four consecutive channels of one pixel: bias, ReLU, round to the storage dtype (the same rounding point
//    as the separate launches; packed adds / conversions), 8-byte write into the LDS image of the stem region; pixels outside the stem
//    map are written as 0 (post-ReLU values are >= 0, so a zero stands in for the pool's padding) -- only tiles on the map's border
//    carry that select.
//  * pool: thread = (pooled pixel, 16 channels): nine 32-byte LDS reads (pieces XOR-swizzled by the column so that neighbouring
//    pooled pixels, two columns apart, do not share banks), maximum on the raw 16-bit patterns (v_pk_max_i16: the values are >= 0),
//    two 16-byte stores; the tile's output rows are 2 KB runs.
// Round 5 rebuilt the kernel from its counters (vector-ALU issue, then LDS reads: DESIGN.md section 3): 222 -> 161 us standalone.
// Same K order (kernel rows ascending, fp32 accumulation) and rounding points as osr_conv2d_fwd(stem view) + osr_maxpool3x3s2.
#include "osr_common.h"
#include <type_traits>

typedef f16_t sp_h8 __attribute__((ext_vector_type(8)));
typedef bf16_t sp_b8 __attribute__((ext_vector_type(8)));
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sp_u32x2 __attribute__((ext_vector_type(2)));

template <class T> struct SpFrag;
template <> struct SpFrag<f16_t> {
    typedef sp_h8 type;
    static __device__ __forceinline__ sp_f32x4 mfma(sp_h8 a, sp_h8 b, sp_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct SpFrag<bf16_t> {
    typedef sp_b8 type;
    static __device__ __forceinline__ sp_f32x4 mfma(sp_b8 a, sp_b8 b, sp_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

#define SP_PH 4                     // pooled rows of a tile
#define SP_PW 16                    // pooled columns
#define SP_SH (2 * SP_PH + 1)       // 9 stem rows
#define SP_SW (2 * SP_PW + 1)       // 33 stem columns
#define SP_NPIX (SP_SH * SP_SW)     // 297
#define SP_NGRP ((SP_NPIX + 15) / 16)  // 19 groups of 16 stem pixels
#define SP_IR (2 * (SP_SH - 1) + 7)    // 23 patch rows
#define SP_IC (2 * (SP_SW - 1) + 8)    // 72 patch pixels per row
#define SP_IPITCH (SP_IC * 8)          // 576 bytes
#define SP_PATCH_BYTES (SP_IR * SP_IPITCH)                 // 13 248
#define SP_TILE_OFF ((SP_PATCH_BYTES + 255) / 256 * 256)   // 13 312
#define SP_NIT ((SP_IR * SP_IC + 255) / 256)                // 7 patch pixels per thread
#define SP_LDS (SP_TILE_OFF + SP_NGRP * 16 * 128)          // + 38 912 = 52 224

struct StemPoolArgs {
    const void* x;       // SRC 0: (n, hd, wd, 4) pre-padded, normalised image in the storage dtype; SRC 1 / 2: (n, 3, h, w) uint8 / float32 raw image
    int ih, iw;          // raw image size (SRC 1 / 2)
    float m0, m1, m2, s0, s1, s2;  // pixel mean / std (SRC 1 / 2): the patch is normalised while it is staged, as osr_preprocess does
    const void* w;       // (64, >= 7, 1, 32): stem view, row stride wrow elements
    const float* bias;   // (64)
    void* out;           // (n, hq, wq, 64)
    int n, hd, wd, hs, ws, hq, wq, tiles_x, tiles_y, wrow, ntiles;
};

template <class T, int SRC>
__global__ __launch_bounds__(256, 3) void stem_pool_kernel(StemPoolArgs a) {
    typedef typename SpFrag<T>::type frag_t;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[SP_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- work split inside the workgroup (round 5): wave w owns the 32 channels of half w & 1 (two 16-channel MFMA row groups) and every
    //      second group of 16 stem pixels (w >> 1: groups 0, 2, .. 18 or 1, 3, .. 17). Until then a wave owned 16 channels over ALL pixels:
    //      every pixel fragment was read from LDS by all four waves, and the counters (scripts/exp_stem_pmc.sh) showed the LDS pipe busy 84 %
    //      of the kernel's cycles, 94 % of that in those reads. Two channel groups per wave halve them, within the register budget of three
    //      waves per SIMD (all four groups per wave: 28 weight fragments + 16 biases, 190 registers).
    //      Weight fragments (A operand): lane (m = lane & 15, kg = lane >> 4) holds w[16 cg + m][ky][8 kg .. 8 kg + 7]; loaded ONCE per
    //      workgroup: the workgroups are persistent (three per CU, each walks ~22 tiles at the bench's size) ----
    const int m16 = lane & 15, kg = lane >> 4;
    const int ch = wid & 1, ph = wid >> 1;
    frag_t wf[2][7];
    float4 b4[2];  // the lane's D rows of group cg are channels 16 cg + 4 kg + 0..3
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cg = 2 * ch + j;
        const T* wp = reinterpret_cast<const T*>(a.w) + (size_t)(16 * cg + m16) * a.wrow * 32 + 8 * kg;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) wf[j][ky] = *reinterpret_cast<const frag_t*>(wp + ky * 32);
        b4[j] = *reinterpret_cast<const float4*>(a.bias + 16 * cg + 4 * kg);
    }
    // Per pixel group, this lane's LDS offsets -- where its B fragment starts in the patch, where its channels go in the stem image -- do not
    // depend on the tile: computed once per workgroup, one packed register per group (the group loop is unrolled). The counters had shown
    // the first bound to be vector-ALU issue (75 M VALU instructions per launch = 64 % of all issue cycles, the MFMAs 31 %): 40 per group
    // and wave -- the pixel's row / column, both addresses, the inside-the-map test, then adds, maxima, conversions, selects, packs one value
    // at a time. Now: three to unpack, and per channel group one address, two packed adds, four maxima, two packed conversions; the
    // inside-the-map selects only on the tiles that touch the map's border.
    constexpr int SP_GPW = (SP_NGRP + 1) / 2;  // 10 group slots per wave (the last one of the odd pixel half is empty)
    unsigned g_rd[SP_GPW];  // patch offset of the lane's B fragment | row << 14 | column << 18 | (stem-image slot swizzle of the lane) << 24
#pragma unroll
    for (int i = 0; i < SP_GPW; ++i) {
        const int grp = ph + 2 * i;
        const int q = 16 * grp + m16, qc = q < SP_NPIX ? q : SP_NPIX - 1;  // (lanes past pixel 296 re-read pixel 296 and write slots the pool never reads)
        const int r = (qc * 1986) >> 16, c = qc - r * 33;                  // qc / 33 for qc < 2048 (1986 = ceil(65536 / 33))
        // B operand: lane (n = lane & 15 -> pixel q, kg) reads patch row 2 r + ky, pixels 2 c + 2 kg, 2 c + 2 kg + 1 (16 bytes).
        // D: four channels (8 bytes) of pixel q go to piece 2 cg + (kg >> 1) at slot piece ^ ((c >> 1) & 7), half kg & 1
        g_rd[i] = (unsigned)((2 * r) * SP_IPITCH + (2 * c + 2 * kg) * 8) | ((unsigned)r << 14) | ((unsigned)c << 18) | ((unsigned)(((kg >> 1) ^ (c >> 1)) & 7) << 24);
    }
    const unsigned wr_lane = (unsigned)(SP_TILE_OFF + m16 * 128 + (kg & 1) * 8);  // + 2048 per pixel group (wave-uniform)

    // XCD-aware walk: workgroup b runs on XCD b % 8 (round-robin dispatch); an XCD takes a contiguous run of tiles (row-major inside an image)
    // and its workgroups walk that run side by side -- the patch rows neighbouring tiles share hit one L2. (A grid of fewer than 8
    // workgroups is one group.)
    const int ngrp = gridDim.x >= 8 ? 8 : 1, wpg = gridDim.x / ngrp;            // (the host launches a multiple of 8 workgroups from 8 on)
    const int xg = blockIdx.x % ngrp, tq = a.ntiles / ngrp, tr = a.ntiles % ngrp;
    const int run0 = xg * tq + (xg < tr ? xg : tr), run_len = tq + (xg < tr ? 1 : 0);
#pragma unroll 1
    for (int tl = blockIdx.x / ngrp; tl < run_len; tl += wpg) {
    const int t = run0 + tl;
    const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, img = t / (a.tiles_x * a.tiles_y);
    const int py0 = ty * SP_PH, px0 = tx * SP_PW;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;   // first stem row / column of the region (-1 on the top / left tiles: the pool's padding)
    const int iy0 = 2 * sy0, ix0 = 2 * sx0;           // first patch row / column in the pre-padded image (stem pixel (y, x) reads rows 2y .. 2y + 6)

    // ---- stage the patch. SRC 1 / 2: straight from the raw NCHW image -- (value - mean) / std rounded to the storage dtype, zero in
    //      the 3-pixel halo, the /32 padding and the 4th channel: the bits osr_preprocess writes, without the 139 MB round trip ----
    if constexpr (SRC != 0) {
        typedef typename std::conditional<SRC == 1, unsigned char, float>::type S;
        const size_t plane = (size_t)a.ih * a.iw;
        const S* src0 = reinterpret_cast<const S*>(a.x) + (size_t)img * 3 * plane;  // (wave-uniform plane bases + one 32-bit offset per pixel:
        const S* src1 = src0 + plane;                                                 //  a per-load 64-bit address costs two registers each)
        const S* src2 = src1 + plane;
        typedef T t4 __attribute__((ext_vector_type(4)));
        const bool unit_std = a.s0 == 1.0f && a.s1 == 1.0f && a.s2 == 1.0f;  // (x / 1.0f == x exactly: both Openset yaml files; skips three divisions per pixel)
        // All of a thread's loads go out before the first is used: the rolled loop waited for its three one-byte loads in every one of its
        // seven trips -- seven dependent round trips to L2 / HBM per tile (round 5: `s_waitcnt vmcnt(0)` inside the loop). Same values, same
        // rounding. A patch that lies inside the raw image as a whole (wave-uniform test) skips the per-pixel border tests.
        const int yb = iy0 - 3, xb = ix0 - 3;
        const bool whole = yb >= 0 && xb >= 0 && yb + SP_IR <= a.ih && xb + SP_IC <= a.iw;
        S raw[SP_NIT][3];
        bool inside[SP_NIT];
#pragma unroll
        for (int it = 0; it < SP_NIT; ++it) {
            const int i = tid + it * 256, r = i / SP_IC, px = i - r * SP_IC;
            const int y = yb + r, x = xb + px;
            inside[it] = (it < SP_NIT - 1 || tid + it * 256 < SP_IR * SP_IC) && (whole || (y >= 0 && y < a.ih && x >= 0 && x < a.iw));
            const unsigned o = inside[it] ? (unsigned)(y * a.iw + x) : 0u;  // (a plane is < 2^31 pixels: checked on the host)
            raw[it][0] = src0[o]; raw[it][1] = src1[o]; raw[it][2] = src2[o];  // (element 0 of each plane when outside: a valid address, value unused)
        }
#pragma unroll
        for (int it = 0; it < SP_NIT; ++it) {
            if (it < SP_NIT - 1 || tid + it * 256 < SP_IR * SP_IC) {
                const int i = tid + it * 256, r = i / SP_IC, px = i - r * SP_IC;
                float v0 = 0.f, v1 = 0.f, v2 = 0.f;
                if (inside[it]) {
                    v0 = (float)raw[it][0] - a.m0; v1 = (float)raw[it][1] - a.m1; v2 = (float)raw[it][2] - a.m2;
                    if (!unit_std) { v0 = v0 / a.s0; v1 = v1 / a.s1; v2 = v2 / a.s2; }
                }
                const t4 o4 = {(T)v0, (T)v1, (T)v2, (T)0.f};
                *reinterpret_cast<t4*>(lds + r * SP_IPITCH + px * 8) = o4;
            }
        }
    } else
    // ---- stage the patch: 23 rows x 36 chunks of 16 bytes, zero outside the pre-padded image ----
    {
        const char* xi = reinterpret_cast<const char*>(a.x) + (size_t)img * a.hd * a.wd * 8;
        for (int i = tid; i < SP_IR * (SP_IC / 2); i += 256) {
            const int r = i / (SP_IC / 2), ch = i - r * (SP_IC / 2);
            const int y = iy0 + r, x = ix0 + 2 * ch;  // (two pixels per chunk; ix0 is even, wd is a multiple of 8: a chunk is inside or outside as a whole)
            sp_u32x4 v = {0u, 0u, 0u, 0u};
            if (y >= 0 && y < a.hd && x >= 0 && x + 1 < a.wd) v = *reinterpret_cast<const sp_u32x4*>(xi + ((size_t)y * a.wd + x) * 8);
            *reinterpret_cast<sp_u32x4*>(lds + r * SP_IPITCH + ch * 16) = v;
        }
    }
    __syncthreads();

    // ---- conv + bias + ReLU -> LDS image of the stem region: pixel q = 33 row + column, 128 bytes, piece j (8 channels) at slot j ^ ((column >> 1) & 7) ----
    unsigned char* tile = lds + SP_TILE_OFF;
    // interior tile: every pixel of the 9 x 33 region lies inside the stem map (all but the top row and the left column of tiles at the bench's size)
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + SP_SH <= a.hs && sx0 + SP_SW <= a.ws;
    {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        typedef T t4 __attribute__((ext_vector_type(4)));
        const f32x4v zv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SP_GPW; ++i) {
            if (i == SP_GPW - 1 && ph + 2 * i >= SP_NGRP) break;  // (wave-uniform: the odd pixel half has nine groups)
            unsigned rd = g_rd[i];
            asm volatile("" : "+v"(rd));  // (opaque: the compiler otherwise hoists the unpacked fields of all ten groups out of the tile loop -- 30 registers, spills)
            const unsigned char* pb = lds + (rd & 0x3fffu);
            frag_t xb[7];
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) xb[ky] = *reinterpret_cast<const frag_t*>(pb + ky * SP_IPITCH);
            sp_f32x4 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)  // (kernel rows ascending per accumulator: the K order of the separate launches)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = SpFrag<T>::mfma(wf[j][ky], xb[ky], acc[j]);
            bool in = true;
            if (!interior) {  // (wave-uniform) pixels outside the stem map are written as 0: the pool's padding
                const int r = (int)((rd >> 14) & 15u), c = (int)((rd >> 18) & 63u);
                const int sy = sy0 + r, sx = sx0 + c;
                in = sy >= 0 && sy < a.hs && sx >= 0 && sx < a.ws;
            }
            unsigned char* tq = lds + wr_lane + (unsigned)(ph + 2 * i) * 2048u;
            const unsigned fk4 = (rd >> 20) & 0x70u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4v bv = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
                f32x4v v = __builtin_elementwise_max((f32x4v)acc[j] + bv, zv);  // fp32 add, fp32 max, one rounding: as the separate launches
                if (!interior) v = in ? v : zv;
                *reinterpret_cast<t4*>(tq + (((unsigned)(2 * ch + j) << 5) ^ fk4)) = __builtin_convertvector(v, t4);
            }
            __builtin_amdgcn_sched_barrier(0);  // (one group's fragments at a time: hoisting the next group's reads costs the third wave per SIMD its registers)
        }
    }
    __syncthreads();

    // ---- 3x3 / s2 max pool out of the LDS image: thread = (pooled pixel, 16 channels). The stem pixels are post-ReLU (>= 0, or a zero
    //      standing in for the pool's padding): non-negative f16 / bf16 values order like their bit patterns read as signed 16-bit integers
    //      (-0.0 = 0x8000 is the smallest of them and loses to the +0 the maximum starts from, as it did to fmaxf's 0.f), so the maximum is
    //      taken on the raw halves, two per v_pk_max_i16 -- 72 packed operations per thread instead of 144 conversions + 144 fp32 maxima ----
    {
        const int pix = tid >> 2, cq = tid & 3;
        const int py = pix >> 4, px = pix & 15;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 m0 = {0, 0, 0, 0, 0, 0, 0, 0}, m1 = m0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int c = 2 * px + dx, q = (2 * py + dy) * 33 + c, f = (c >> 1) & 7;
                m0 = __builtin_elementwise_max(m0, *reinterpret_cast<const s16x8*>(tile + q * 128 + (((2 * cq) ^ f) << 4)));
                m1 = __builtin_elementwise_max(m1, *reinterpret_cast<const s16x8*>(tile + q * 128 + (((2 * cq + 1) ^ f) << 4)));
            }
        const int oy = py0 + py, ox = px0 + px;
        if (oy < a.hq && ox < a.wq) {
            T* op = reinterpret_cast<T*>(a.out) + (((size_t)img * a.hq + oy) * a.wq + ox) * 64 + 16 * cq;
            *reinterpret_cast<s16x8*>(op) = m0;
            *reinterpret_cast<s16x8*>(op + 8) = m1;
        }
    }
    // (no barrier here: the next tile's staging writes the PATCH, which nobody reads after the barrier in front of the pool; its conv writes
    // the stem image only behind the next barrier, which every wave reaches after its pool reads)
    }
}

static osr_status stem_pool_launch(StemPoolArgs& a, int32_t n, int32_t hp, int32_t wp, int32_t w_rows, int32_t dtype, int src_kind, hipStream_t st) {
    a.n = n; a.hd = hp + 6; a.wd = osr_stem_padded_width(wp);
    a.hs = hp / 2; a.ws = wp / 2;                         // stem output: floor((hp + 6 - 7) / 2) + 1
    a.hq = (a.hs - 1) / 2 + 1; a.wq = (a.ws - 1) / 2 + 1;  // pool output
    a.tiles_x = (a.wq + SP_PW - 1) / SP_PW; a.tiles_y = (a.hq + SP_PH - 1) / SP_PH;
    a.wrow = w_rows;
    const long long tiles = (long long)n * a.tiles_x * a.tiles_y;
    OSR_REQUIRE(tiles < (1ll << 31), OSR_ERR_UNSUPPORTED, "osr_stem_maxpool_fwd: too many tiles");
    a.ntiles = (int)tiles;
    long long g = tiles < 768 ? tiles : 768;  // persistent workgroups: three per CU (51 KB of LDS each) on 256 CUs
    if (g >= 8) g &= ~7ll;
    const dim3 grid((unsigned)g), block(256);
#define SP_LAUNCH(T, K) hipLaunchKernelGGL((stem_pool_kernel<T, K>), grid, block, 0, st, a)
    if (dtype == OSR_F16) { if (src_kind == 0) SP_LAUNCH(f16_t, 0); else if (src_kind == 1) SP_LAUNCH(f16_t, 1); else SP_LAUNCH(f16_t, 2); }
    else                  { if (src_kind == 0) SP_LAUNCH(bf16_t, 0); else if (src_kind == 1) SP_LAUNCH(bf16_t, 1); else SP_LAUNCH(bf16_t, 2); }
#undef SP_LAUNCH
    OSR_CHECK_LAUNCH("osr_stem_maxpool_fwd");
    return OSR_OK;
}

extern "C" osr_status osr_stem_maxpool_fwd(const void* xpad, int32_t n, int32_t hp, int32_t wp, const void* w_view, int32_t w_rows, const float* bias,
                                           void* out, int32_t dtype, void* stream) {
    OSR_REQUIRE(xpad && w_view && bias && out, OSR_ERR_INVALID_ARG, "osr_stem_maxpool_fwd: null pointer");
    OSR_REQUIRE(dtype == OSR_F16 || dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_stem_maxpool_fwd: f16 / bf16 storage");
    OSR_REQUIRE(n >= 1 && hp >= 2 && wp >= 2 && hp % 2 == 0 && wp % 2 == 0 && (w_rows == 7 || w_rows == 8), OSR_ERR_INVALID_ARG,
                "osr_stem_maxpool_fwd: bad sizes (even padded image, 7- or 8-row stem view)");
    OSR_REQUIRE((((uintptr_t)xpad | (uintptr_t)w_view | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_stem_maxpool_fwd: pointers must be 16-byte aligned");
    StemPoolArgs a;
    a.x = xpad; a.w = w_view; a.bias = bias; a.out = out;
    a.ih = a.iw = 0; a.m0 = a.m1 = a.m2 = 0.f; a.s0 = a.s1 = a.s2 = 1.f;
    return stem_pool_launch(a, n, hp, wp, w_rows, dtype, 0, (hipStream_t)stream);
}

// The same from the RAW image: osr_preprocess ([d2] GeneralizedRCNN.preprocess_image + ImageList.from_tensors) folded into the staging.
extern "C" osr_status osr_stem_maxpool_fwd_raw(const void* src, int32_t src_is_u8, int32_t n, int32_t h, int32_t w, int32_t hp, int32_t wp,
                                               const float mean[3], const float stdv[3], const void* w_view, int32_t w_rows, const float* bias,
                                               void* out, int32_t dtype, void* stream) {
    OSR_REQUIRE(src && mean && stdv && w_view && bias && out, OSR_ERR_INVALID_ARG, "osr_stem_maxpool_fwd_raw: null pointer");
    OSR_REQUIRE(dtype == OSR_F16 || dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_stem_maxpool_fwd_raw: f16 / bf16 storage");
    OSR_REQUIRE(n >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w && hp % 2 == 0 && wp % 2 == 0 && (w_rows == 7 || w_rows == 8), OSR_ERR_INVALID_ARG,
                "osr_stem_maxpool_fwd_raw: bad sizes (even padded size >= image size, 7- or 8-row stem view)");
    OSR_REQUIRE((((uintptr_t)w_view | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_stem_maxpool_fwd_raw: pointers must be 16-byte aligned");
    StemPoolArgs a;
    a.x = src; a.w = w_view; a.bias = bias; a.out = out;
    a.ih = h; a.iw = w; a.m0 = mean[0]; a.m1 = mean[1]; a.m2 = mean[2]; a.s0 = stdv[0]; a.s1 = stdv[1]; a.s2 = stdv[2];
    return stem_pool_launch(a, n, hp, wp, w_rows, dtype, src_is_u8 ? 1 : 2, (hipStream_t)stream);
}
