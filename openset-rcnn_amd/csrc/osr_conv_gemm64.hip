// Fast path of osr_conv2d_fwd for cin % 64 == 0 (and the 8-tap stem view): BK = 64 implicit GEMM with
// direct-to-LDS loads (buffer_load ... lds, 16 B per lane, 1 KiB per wave-instruction).
//
//  * LDS tile rows are 128 B (64 halves) with NO padding -- an LDS-DMA wave-instruction writes 1 KiB linearly
//    (8 rows x 8 chunks) -- and bank conflicts are removed by an XOR swizzle applied on the SOURCE side: the
//    lane that lands in slot s of row r fetches logical chunk  s ^ ((r >> 1) & 7); the MFMA fragment read of
//    logical chunk c from row r goes to slot  c ^ ((r >> 1) & 7)  (conflict-free for the ds_read_b128 lane groups).
//  * zero padding / tile tails cost nothing: invalid lanes get a buffer offset beyond num_records and the
//    buffer bounds check writes zeros into LDS.
//  * one barrier per K step: [vmcnt(0)] -> barrier -> issue tile k+1 -> 16 MFMA (32x32x16) per wave on tile k.
//  * workgroup -> tile mapping is XCD-aware (blocks b and b+8 share an XCD/L2): each XCD walks a contiguous run of
//    tiles with the N tiles of one M tile adjacent, so the gathered A rows are fetched into one L2 once.
#include "osr_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Fixed design choices, each measured against its alternative on one box (profiles/README.md):
//  * the wait in front of the first K step waits for everything in flight (a counted wait that lets the bias / residual
//    prefetch stay in flight was 6 % slower on the family);
//  * v_mfma_f32_16x16x32 (four per 32x32 macro tile and 32-wide K step), 5-8 % faster here than v_mfma_f32_32x32x16;
//  * two fragment register sets in every variant (a single set saved 32 VGPRs and nothing else).
#ifndef C64_SINGLE_MINW
#define C64_SINGLE_MINW 2  // waves per SIMD the single-buffer 128-wide kernels must allow (4 spills)
#endif
#ifndef C64_PSPREAD
#define C64_PSPREAD 2      // the staging pieces of the next K slice go out during the first 1/C64_PSPREAD of the MFMAs
#endif
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

template <class T> struct Frag64;
template <> struct Frag64<f16_t> {
    typedef f16x8 type;
    static __device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Frag64<bf16_t> {
    typedef bf16x8 type;
    static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// Division of a 32-bit row index by an invariant (Granlund-Montgomery, unsigned): the magic pair is computed on the host,
// the device pays one mul_hi and four integer ops instead of the ~150-instruction 64-bit division the row -> (n, oh, ow)
// decomposition otherwise costs (it runs per staged row in the prologue and per stored row segment in the epilogue; stamped
// diagnostics showed it, not memory, was most of a short-K tile's prologue and epilogue).
struct FastDiv {
    unsigned mp, sh1, sh2, d;
};
static FastDiv fastdiv_make(unsigned d) {
    FastDiv f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;  // ceil(log2 d)
    f.mp = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    f.d = d;
    return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
    const unsigned t = __umulhi(f.mp, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
// row index -> image, output row, output column
#define C64_ROW_TO_NHW(m_, nimg_, oh_, ow_)                          \
    const unsigned mu_##nimg_ = (unsigned)(m_);                      \
    const int nimg_ = (int)fastdiv(mu_##nimg_, e_div_howo);          \
    const unsigned rem_##nimg_ = mu_##nimg_ - (unsigned)nimg_ * e_div_howo.d; \
    const int oh_ = (int)fastdiv(rem_##nimg_, e_div_wo);             \
    const int ow_ = (int)(rem_##nimg_ - (unsigned)oh_ * e_div_wo.d);

#define C64_MAX_LEVELS 6
#define C64_TW_LD 264  // elements per row of the fused CF-RPN tail's split weight matrix in LDS (528 B)
// bytes of LDS behind the parked tile: the (16, C64_TW_LD) weight matrix, later overwritten by the (rows, 16) fp32 slab of the tail's accumulators
// biased fp32 exponent of a tail row's largest |weight| (bit pattern b), kept in [11, 254] so that both 2^(10 - e) and 2^(e - 10) are normal floats
#define C64_TAIL_EXP(b) (min(max((b) >> 23, 11u), 254u))
#define C64_TAIL_LDS(bm) ((size_t)(16 * C64_TW_LD * 2) > (size_t)(bm) * 64 ? (size_t)(16 * C64_TW_LD * 2) : (size_t)(bm) * 64)
static_assert(C64_MAX_LEVELS == OSR_MAX_CONV_LEVELS, "include/osr.h: OSR_MAX_CONV_LEVELS");
struct Conv64Args {
    osr_conv_params p;
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    const void* mask;      // nullable: ReLU mask applied after bias / residual (osr_conv2d_fwd_masked), out's layout, in_dtype
    void* out;
    long long M;
    int K;                 // GEMM K of the weight rows (multiple of 64)
    int tiles_m, tiles_n;
    unsigned in_bytes, w_bytes;  // buffer sizes for the bounds check (< 2 GiB)
    int stem;              // 1: cin == 32 view, two taps per K slice
    int two_stage;         // 1: double-buffered staging (K-heavy layers); 0: one staging buffer, more workgroups per CU
    int tap_minor;         // 1: K runs channel-slice-major / tap-minor (L2-friendly for KH*KW > 1), 0: tap-major
    int pw_dense = 0;      // dense-row shortcuts (row m of a dense (rows, channels) tensor starts at m * channels, < 2^31 elements): the row -> (image, y, x)
                           // decomposition and the strided 64-bit addresses drop out. bit 0: the INPUT side (1 x 1, stride 1, no padding, dense input:
                           // the gather descriptors of the prologue); bit 1: dense OUTPUT (and ReLU mask) rows; bit 2: dense RESIDUAL rows (modes 1 / 3)
    // fused CF-RPN tail (EPI == 1): 1x1 weights [5][256] (rows 0-3 ltrb deltas, row 4 centerness), biases, outputs
    const float* tail_w;
    const float* tail_b;
    float* tail_deltas;
    float* tail_ctr;
    int tail_lds_off;      // byte offset of the tail weights in LDS
    FastDiv div_howo, div_wo;  // row -> (image, oh, ow); M < 2^31
    // Split-K of the last, partial dispatch round (deep-K 1x1 / FC layers; conv64_launch_split): a launch covers the linear tiles
    // [tile0, tile0 + ntile) only; with ksplit > 1 every one of them is taken by ksplit workgroups, each over its own range of K
    // slices, which write raw fp32 partial sums (no bias / ReLU) into slab s = out + s * split_stride floats.
    int tile0, ntile, ksplit;
    long long split_stride;
    // chained 1x1 convolution (EPI == 2, osr_conv2d_chain_fwd): out = relu(conv1x1(relu(conv(in) + bias), w3) + bias3 + res)
    const void* w3;        // [cout3][cout] packed, storage dtype
    const float* bias3;
    int cout3;
    unsigned w3_bytes, out_bytes;  // buffer sizes of w3 and of out / res (dense rows of cout3 elements)
    void* mid_out;         // nullable: the parked relu(conv + bias) tile also goes to HBM, dense (rows, 128): the training step keeps it for the backward
    // Multi-level launch (osr_conv2d_fwd_levels / osr_cfrpn_head_fwd_levels: the 256 x 256 8-phase instantiations only): the same
    // convolution over several dense NHWC inputs -- the FPN output convs, the CF-RPN head over p2..p6 -- as ONE grid. The M tiles of
    // level l are the linear tiles [lv[l].tile_begin, lv[l + 1].tile_begin); a workgroup looks its level up and takes the fields below
    // from it instead of from the single-problem fields above. nlevels == 0: a single problem.
    // Column groups (osr_conv2d_fwd_pair: EPI == 0, SPLIT == 0 instantiations): TWO convolutions of the same input and geometry -- the
    // projection shortcut and conv1 of a stage's first bottleneck -- as one grid: the N tiles [cg[g].tile_begin, ...) belong to group g,
    // which brings its own weights, bias, ReLU flag and dense output. ngroups <= 1: a single convolution.
    int ngroups = 0;
    struct Group {
        const void* w; const float* bias; void* out;
        long long out_stride_n, out_stride_h;
        int tile_begin, cout, relu;
        unsigned w_bytes;
    } cg[2];
    int nlevels = 0;
    struct Level {
        const void* in; void* out; float* tail_deltas; float* tail_ctr;
        const void* w; const float* bias;  // the level's own weights / bias (the FPN's output convs), or the shared ones
        long long M, in_stride_n, in_stride_h, out_stride_n, out_stride_h;
        int tile_begin, hi, wi;
        unsigned in_bytes;
        FastDiv div_howo, div_wo;
    } lv[C64_MAX_LEVELS];
#ifdef C64_STAMPS
    unsigned long long* dbg;
    unsigned long long* p8;
#endif
};

#define OOB_OFF 0x80000000u

// Diagnostic build only (-DC64_STAMPS, never shipped): wave 0 of every workgroup records s_memrealtime (100 MHz) at
// kernel entry, after the first staged slice has landed, after the K loop and at exit into a caller-provided buffer.
#ifdef C64_STAMPS
static unsigned long long* g_c64_stamps = nullptr;
extern "C" void osr_debug_set_conv_stamps(unsigned long long* p) { g_c64_stamps = p; }
#define C64_STAMP(i) if (a.dbg && tid == 0) a.dbg[(long long)blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime()
// 8-phase K loop: shader-clock stamps of K tile P8_STAMP_TILE at the section boundaries of its four phases (phase start, fragment
// reads issued, pieces issued, first barrier passed, MFMAs issued), kept in registers and written after the loop (a store inside the
// loop would count on vmcnt); lanes 0 of waves 0 and 4 -> p8[(block * 2 + wave group) * 24 + i]
static unsigned long long* g_p8_stamps = nullptr;
extern "C" void osr_debug_set_p8_stamps(unsigned long long* p) { g_p8_stamps = p; }
#define P8_STAMP_TILE 6
#define P8_STAMP(i) if (ks == P8_STAMP_TILE) p8s[i] = __builtin_amdgcn_s_memtime()
#else
#define C64_STAMP(i)
#define P8_STAMP(i)
#endif

template <class TO> __device__ __forceinline__ void store8_64(TO* p, const float v[8]);
template <> __device__ __forceinline__ void store8_64<float>(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8_64<f16_t>(f16_t* p, const float v[8]) {
    f16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (f16_t)v[i];
    *reinterpret_cast<f16x8*>(p) = t;
}
template <> __device__ __forceinline__ void store8_64<bf16_t>(bf16_t* p, const float v[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned c64_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int c64_chain_perm(int row) { return (row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3); }
template <int N> __device__ __forceinline__ void c64_wait_vm_lgkm_barrier() {  // ONE statement: the compiler may not put anything between the waits and the barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void c64_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SPLIT = 1: the split-K tail launch (fp32 partial sums of a K range per workgroup); a template parameter so that the ordinary
// instantiations carry none of its registers (four more VGPRs cost the 128 x 128 single-buffer kernel its third wave per SIMD).
template <class TI, class TO, int BM, int BN, int WM, int WN, int EPI, int TWO, int SPLIT = 0>
__global__ __launch_bounds__(WM * WN * 64, (BM * BN == 256 * 256 && WM * WN == 4) ? 1 : (TWO || BM * BN > 128 * 128 || EPI != 0) ? 2 : C64_SINGLE_MINW) void conv_igemm64_kernel(Conv64Args a) {  // waves per SIMD the register budget must allow
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NW = WM * WN, NT = NW * 64;  // waves / threads per workgroup (4 or 8 waves)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int A_PIECES = BM / 8 / NW, B_PIECES = BN / 8 / NW;  // 1-KiB LDS-DMA pieces per wave per K step
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(TN % 2 == 0, "epilogue works on pairs of 32-column tiles");
    constexpr int EPI_LD = 64 + 4;  // floats per staged row: one pair of N tiles at a time
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // max(stages * STAGE, EPI_BYTES), see conv64_launch

    typedef typename Frag64<TI>::type frag_t;
    constexpr bool PH8 = (TWO == 2);  // the 8-phase K loop (below)
    const osr_conv_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;
    C64_STAMP(0);

    // XCD-aware bijective remap of the linear block id (guide T1)
    const int nwg = gridDim.x;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int ksp = 0;  // which K range of the tile (split-K tail launches only)
    if constexpr (SPLIT) { ksp = t / a.ntile; t = a.tile0 + t % a.ntile; }
    const int tile_n = t % a.tiles_n;
    int tile_m = t / a.tiles_n;
    // the level-dependent fields of the problem (wave-uniform: scalar registers)
    long long e_M = a.M, e_isn = p.in_stride_n, e_ish = p.in_stride_h, e_osn = p.out_stride_n, e_osh = p.out_stride_h;
    const void* e_in = a.in;
    const void* e_w = a.w;
    const float* e_bias = a.bias;
    void* e_out = a.out;
    float* e_tail_deltas = a.tail_deltas;
    float* e_tail_ctr = a.tail_ctr;
    int e_hi = p.hi, e_wi = p.wi;
    unsigned e_in_bytes = a.in_bytes;
    FastDiv e_div_howo = a.div_howo, e_div_wo = a.div_wo;
    if constexpr (PH8 && SPLIT == 0) {
        if (a.nlevels > 0) {
            int l = 0;
            for (int i = 1; i < a.nlevels; ++i) l = tile_m >= a.lv[i].tile_begin ? i : l;
            const Conv64Args::Level& L = a.lv[l];
            e_M = L.M; e_isn = L.in_stride_n; e_ish = L.in_stride_h; e_osn = L.out_stride_n; e_osh = L.out_stride_h;
            e_in = L.in; e_w = L.w; e_bias = L.bias; e_out = L.out; e_tail_deltas = L.tail_deltas; e_tail_ctr = L.tail_ctr;
            e_hi = L.hi; e_wi = L.wi; e_in_bytes = L.in_bytes; e_div_howo = L.div_howo; e_div_wo = L.div_wo;
            tile_m -= L.tile_begin;
        }
    }
    int e_cout = p.cout, e_relu = p.relu, tile_n_l = tile_n;
    long long e_osw = p.out_stride_w;
    unsigned e_w_bytes = a.w_bytes;
    if constexpr (EPI == 0 && SPLIT == 0) {
        if (a.ngroups > 1) {
            const Conv64Args::Group& G = a.cg[tile_n >= a.cg[1].tile_begin ? 1 : 0];
            e_w = G.w; e_bias = G.bias; e_out = G.out; e_osn = G.out_stride_n; e_osh = G.out_stride_h; e_osw = G.cout;
            e_cout = G.cout; e_relu = G.relu; e_w_bytes = G.w_bytes;
            tile_n_l = tile_n - G.tile_begin;
        }
    }
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n_l * BN;  // (first output channel of the tile, inside its column group)
    if (p.row_seg_counts) {  // segmented rows (padded per-image lists): a tile without a single data row has nothing to do (wave-uniform)
        const long long sr = p.row_seg_rows, mend = m0 + BM < e_M ? m0 + BM : e_M;
        bool any = false;
        for (long long sg = m0 / sr; sg * sr < mend; ++sg) {
            const long long lo = m0 > sg * sr ? m0 : sg * sr, hi = sg * sr + p.row_seg_counts[sg];
            any |= (hi < mend ? hi : mend) > lo;
        }
        if (!any) return;
    }

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(e_in), 0, e_in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(e_w), 0, e_w_bytes, 0x00020000);

    // ---- per-lane gather descriptors: this lane serves row (piece*8 + lane/8), LDS slot lane%8.
    //      a_off0 = byte offset of element (n, ih0, iw0, chunk) -- wraps below zero where the window starts in the padding --
    //      and a_mask has bit t set when tap t = kh*KW + kw of this row lies inside the image. The K loop then needs one add
    //      and one select per piece: offset = mask bit ? a_off0 + (scalar tap offset) : out of bounds (zero fill).
    //      Bit 31 is never set: tap 31 is the "issue nothing" tap of the last K step. ----
    const int lrow = lane >> 3, slot = lane & 7;
    // TWO == 2 (the 8-phase K loop below): the 256 x 256 tile is staged in four UNITS of 16 pieces, the rows that one phase's fragment reads
    // cover: activation rows of row half qa of both wave rows, weight rows of column half qb of all four wave columns; a wave issues
    // two pieces of every unit (descriptor j = 2 q + jj). Otherwise a wave's pieces are consecutive.
    static_assert(!PH8 || (BM == 256 && BN == 256 && WM == 2 && WN == 4 && EPI != 2), "8-phase K loop: 256 x 256 tile, 2 x 4 waves");
    auto a_piece = [&](int j) -> int {
        if constexpr (PH8) { const int lp = wid * 2 + (j & 1); return (lp >> 3) * 16 + (j >> 1) * 8 + (lp & 7); }
        else return wid * A_PIECES + j;
    };
    auto b_piece = [&](int j) -> int {
        if constexpr (PH8) { const int lp = wid * 2 + (j & 1); return (lp >> 2) * 8 + (j >> 1) * 4 + (lp & 3); }
        else return wid * B_PIECES + j;
    };
    unsigned a_off0[A_PIECES], a_mask[A_PIECES];
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int row = a_piece(j) * 8 + lrow;
        const long long m = m0 + row;
        const bool ok = m < e_M;
        const int chunk = slot ^ ((row >> 1) & 7);
        if (a.pw_dense & 1) {  // (wave-uniform) a dense 1 x 1 layer: row m starts at m * cin, its only tap is always inside
            a_off0[j] = ok ? ((unsigned)m * (unsigned)p.cin + (unsigned)chunk * 8u) * 2u : 0u;
            a_mask[j] = ok ? 1u : 0u;
            continue;
        }
        const long long mm = ok ? m : 0;
        C64_ROW_TO_NHW(mm, nimg, oh, ow);
        const int ih0 = oh * p.stride_h - p.pad_h, iw0 = ow * p.stride_w - p.pad_w;
        // stem view: two 32-wide taps per K slice, chunks 0-3 -> tap kh, 4-7 -> tap kh+1
        const long long coff = a.stem ? (long long)(chunk >> 2) * e_ish + (chunk & 3) * 8 : (long long)chunk * 8;
        a_off0[j] = (unsigned)(((long long)nimg * e_isn + (long long)ih0 * e_ish + (long long)iw0 * p.in_stride_w + coff) * 2);
        unsigned mk = 0x7fffffffu;
        if (p.pad_mode == 0) {
            const int wlo = iw0 < 0 ? -iw0 : 0, whi = e_wi - iw0 < p.kw ? e_wi - iw0 : p.kw;
            const unsigned wm = whi > wlo ? ((1u << whi) - 1u) & ~((1u << wlo) - 1u) : 0u;
            mk = 0u;
            for (int t = 0; t < p.kh; ++t)
                if ((unsigned)(ih0 + t) < (unsigned)e_hi) mk |= wm << (t * p.kw);
        }
        a_mask[j] = ok ? mk : 0u;
    }
    unsigned b_off[B_PIECES];
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) {
        const int row = b_piece(j) * 8 + lrow;
        const unsigned chunk = (unsigned)(slot ^ ((row >> 1) & 7));
        // EPI == 2 runs the MFMAs with the weights as the A operand (D = W X^T: a lane then holds output channels of ONE pixel); LDS row
        // rho of a 32-row group takes weight row 8 ((rho >> 2) & 3) + 4 (rho >> 4) + (rho & 3), so that the lane's rows 4g..4g+3 of the
        // two 16-row sub-tiles are the eight consecutive channels 8g..8g+7 (one 16-byte chunk of the parked image / of the output)
        const int wrow = EPI == 2 ? c64_chain_perm(row) : row;
        const long long o = ((long long)(n0 + wrow) * a.K + chunk * 8) * 2;
        b_off[j] = (n0 + wrow < e_cout && o < (long long)OOB_OFF) ? (unsigned)o : OOB_OFF;
    }

    int kh = 0, kw = 0, c0 = 0;  // tap / channel origin of the current K slice (non-stem)
    int kbyte = 0;               // byte offset of the K slice inside a weight row (scalar offset of the B loads)
    int tap = 0;                 // kh*KW + kw (0 for the stem view, whose taps are never padded); 31 = issue nothing
    unsigned tap_off = 0;        // byte offset of (kh, kw, c0) relative to the window origin
    // one 1-KiB LDS-DMA piece: q < A_PIECES -> activation rows, else weight rows
#define C64_ISSUE_PIECE(stage, q)                                                                                                 \
    {                                                                                                                             \
        unsigned char* sa_ = lds + (stage) * STAGE;                                                                               \
        if ((q) < A_PIECES) {                                                                                                     \
            const int j_ = (q) < A_PIECES ? (q) : 0;                                                                              \
            const unsigned off_ = ((a_mask[j_] >> tap) & 1u) ? a_off0[j_] + tap_off : OOB_OFF;                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t*)(sa_ + a_piece(j_) * 1024), 16, off_, 0, 0, 0);            \
        } else {                                                                                                                  \
            const int j_ = (q) >= A_PIECES ? (q) - A_PIECES : 0;                                                                  \
            const unsigned boff_ = b_off[j_]; /* plain variables only: a type-dependent argument makes the host pass drop the kernel */ \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(sa_ + BM * 128 + b_piece(j_) * 1024), 16, boff_, kbyte, 0, 0); \
        }                                                                                                                         \
    }
#define C64_ISSUE(stage)                                                                  \
    {                                                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < A_PIECES + B_PIECES; ++q_) C64_ISSUE_PIECE(stage, q_); \
    }
    // K order of a KH x KW conv: channel slice outermost, taps innermost (a.tap_minor). Consecutive K steps then re-read the
    // same 64 channels of neighbouring pixels, so eight of the nine tap loads of a 3x3 conv hit the XCD's L2 instead of
    // the Infinity Cache (tap-major order re-reads a pixel only after a full sweep of the channels, ~4 MB of other traffic
    // per XCD later). The weight rows stay [kh][kw][cin]: only the scalar offset sequence of the B loads changes.
#define C64_ADVANCE()                                                                                 \
    {                                                                                                 \
        if (a.stem) { kbyte += 128; kh += 2; tap_off = (unsigned)(kh * e_ish * 2); }          \
        else if (a.tap_minor) {                                                                       \
            if (++kw >= p.kw) { kw = 0; if (++kh >= p.kh) { kh = 0; c0 += 64; } }                     \
            tap = kh * p.kw + kw;                                                                     \
            tap_off = (unsigned)((kh * e_ish + kw * p.in_stride_w + c0) * 2);                 \
            kbyte = (tap * p.cin + c0) * 2;                                                           \
        } else {                                                                                      \
            kbyte += 128;                                                                             \
            c0 += 64;                                                                                 \
            if (c0 >= p.cin) { c0 = 0; if (++kw >= p.kw) { kw = 0; ++kh; } }                          \
            tap = kh * p.kw + kw;                                                                     \
            tap_off = (unsigned)((kh * e_ish + kw * p.in_stride_w + c0) * 2);                 \
        }                                                                                             \
    }

    // Accumulators of the wave's TM x TN macro tiles of 32 x 32, each four 16x16 sub-tiles [si][sj] (f32x4): element q sits at
    // row si*16 + (lane>>4)*4 + q, column sj*16 + (lane&15). C64_ACC(i, j, r), r = 0..15, addresses them uniformly.
    f32x4 acc[TM][TN][2][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][(r >> 3) & 1][(r >> 2) & 1][r & 3] = 0.f;
#define C64_ACC(i, j, r) acc[i][j][((r) >> 3) & 1][((r) >> 2) & 1][(r) & 3]
#define C64_ROW(r) ((((r) >> 3) & 1) * 16 + (lane >> 4) * 4 + ((r) & 3))
#define C64_COL(r) ((((r) >> 2) & 1) * 16 + (lane & 15))

    int nk = a.K / 64;
    if constexpr (SPLIT) {  // this workgroup's share of the K slices: the K state of slice k_lo in the launch's K order
        const int k_lo = (int)((long long)ksp * nk / a.ksplit), k_hi = (int)((long long)(ksp + 1) * nk / a.ksplit);
        const int taps = p.kh * p.kw, spt = p.cin / 64;  // taps; channel slices per tap
        const int tp = a.tap_minor ? k_lo % taps : k_lo / spt, cs = a.tap_minor ? k_lo / taps : k_lo % spt;
        kh = tp / p.kw; kw = tp - kh * p.kw; c0 = cs * 64; tap = tp;
        tap_off = (unsigned)((kh * e_ish + kw * p.in_stride_w + c0) * 2);
        kbyte = (tap * p.cin + c0) * 2;
        nk = k_hi - k_lo;
    }
    C64_ISSUE(0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue operands that do not depend on the accumulators are fetched now, under the K loop:
    //      this lane's bias values and (128x128 / 128x64 tiles) its residual segments, one 16-byte load per staged row ----
    constexpr int RPP = 8, NPASS = 4;               // 8 lanes x 8 channels cover the 64 staged columns; 8 rows per pass
    constexpr int TNP = TN / 2;                     // pairs of N tiles
    constexpr bool PRE_RES = (TN == 2) && (TM <= 2) && (EPI == 0);
    const int cseg = (lane & 7) * 8;
    const TI* __restrict__ res = reinterpret_cast<const TI*>(a.res);
    float bias8[TNP][8];
    frag_t rres[PRE_RES ? TM : 1][PRE_RES ? NPASS : 1];
    if constexpr (EPI == 0) {
#pragma unroll
        for (int jp = 0; jp < TNP; ++jp) {
            const int co = n0 + (wc * TN + jp * 2) * 32 + cseg;
            const int cb = co < e_cout ? co : 0;
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b0 = SPLIT ? zero4 : *reinterpret_cast<const float4*>(e_bias + cb);
            const float4 b1 = SPLIT ? zero4 : *reinterpret_cast<const float4*>(e_bias + cb + 4);
            bias8[jp][0] = b0.x; bias8[jp][1] = b0.y; bias8[jp][2] = b0.z; bias8[jp][3] = b0.w;
            bias8[jp][4] = b1.x; bias8[jp][5] = b1.y; bias8[jp][6] = b1.z; bias8[jp][7] = b1.w;
        }
        if constexpr (PRE_RES) {
            if (p.res_mode != 0) {
                const int co = n0 + wc * TN * 32 + cseg;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pass = 0; pass < NPASS; ++pass) {
                        const long long m = m0 + (wr * TM + i) * 32 + pass * RPP + (lane >> 3);
                        const bool ok = m < e_M && co < e_cout;
                        if (a.pw_dense & 4) {  // (wave-uniform) dense rows: the residual of row m starts at m * cout
                            rres[i][pass] = *reinterpret_cast<const frag_t*>(res + (ok ? (unsigned)m * (unsigned)e_cout + (unsigned)co : 0u));
                            continue;
                        }
                        const long long mm = ok ? m : 0;
                        C64_ROW_TO_NHW(mm, nimg, oh, ow);
                        const int rh = p.res_mode == 2 ? (oh >> 1) : oh, rw = p.res_mode == 2 ? (ow >> 1) : ow;
                        const long long off = ok ? (long long)nimg * p.res_stride_n + (long long)rh * p.res_stride_h + (long long)rw * p.res_stride_w + co : 0ll;
                        rres[i][pass] = *reinterpret_cast<const frag_t*>(res + off);
                    }
            }
        }
    }
    float cb2[EPI == 2 ? TN : 1][8];  // EPI == 2: the conv bias of this lane's eight consecutive channels per 32-channel group
    if constexpr (EPI == 2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* bp = e_bias + (wc * TN + j) * 32 + (lane >> 4) * 8;
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
            cb2[j][0] = b0.x; cb2[j][1] = b0.y; cb2[j][2] = b0.z; cb2[j][3] = b0.w;
            cb2[j][4] = b1.x; cb2[j][5] = b1.y; cb2[j][6] = b1.z; cb2[j][7] = b1.w;
        }
    }
    float tbias[EPI == 1 ? TN : 1][2];  // per-lane conv bias of its output columns ([.][1] only differs for 16x16 sub-tiles)
    if constexpr (EPI == 1) {
        // tail weights -> LDS (behind the staging / t-tile region); per-lane conv bias of its TN output columns
        // The fp32 tail weights as THREE storage-dtype terms each (w * 2^s = hi + mid + lo) -> a (16, 256) B operand for the tail's MFMAs:
        // rows q, 5 + q, 10 + q = hi, mid, lo of tail row q; row 15 zero. Row pitch C64_TW_LD elements (528 B: conflict-free fragment reads).
        // Each tail row is first scaled by a power of two that brings its largest weight to [2^10, 2^11) (exact; undone on the fp32 sums
        // in the epilogue): bf16's three 8-bit terms cover fp32's 24 significand bits at any magnitude, but fp16's terms run out of
        // EXPONENT first -- at this head's initial scale (std 0.01) `mid` was an fp16 subnormal and `lo` underflowed, a per-weight error
        // of ~3e-6 relative that grew as weight decay shrank the row. Scaled, every weight down to 2^-11 of its row's
        // largest splits exactly (lo >= 2^-24 = fp16's subnormal spacing; the MFMA takes subnormal inputs as they are:
        // tests/test_conv_8phase.py, tiny-weight case), and a smaller one is off by at most 2^-35 of the row's largest.
        TI* s_w16 = reinterpret_cast<TI*>(lds + a.tail_lds_off);
        unsigned int* s_tmax = reinterpret_cast<unsigned int*>(lds + a.tail_lds_off + C64_TAIL_LDS(BM));  // bit patterns of max |w| per tail row
        if (tid < 8) s_tmax[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < 5 * 256; i += NT) {  // (a wave's 64 consecutive i lie in one row; trip counts are wave-uniform)
            unsigned int b = __float_as_uint(fabsf(a.tail_w[i]));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) b = max(b, (unsigned int)__shfl_xor((int)b, o, 64));
            if (lane == 0) atomicMax(&s_tmax[i >> 8], b);
        }
        __syncthreads();
        for (int i = tid; i < 5 * 256; i += NT) {
            const int q = i >> 8, k = i & 255;
            const float w = a.tail_w[i] * __uint_as_float((264u - C64_TAIL_EXP(s_tmax[q])) << 23);  // x 2^(10 - exponent of the row's max)
            const TI hi = (TI)w;
            const float r1 = w - (float)hi;
            const TI mid = (TI)r1;
            const TI lo = (TI)(r1 - (float)mid);
            s_w16[q * C64_TW_LD + k] = hi; s_w16[(5 + q) * C64_TW_LD + k] = mid; s_w16[(10 + q) * C64_TW_LD + k] = lo;
        }
        for (int i = tid; i < 256; i += NT) s_w16[15 * C64_TW_LD + i] = (TI)0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            tbias[j][0] = e_bias[(wc * TN + j) * 32 + C64_COL(0)];
            tbias[j][1] = e_bias[(wc * TN + j) * 32 + C64_COL(4)];
        }
    }
    // 16x16x32: a fragment = 16 rows x 32 K; lane l holds row l&15, K chunk (l>>4) of the 32-wide step.
    // [set][tile][row half]; two 32-wide K steps per 64-wide slice, the second one's fragments are read under the first one's MFMAs.
#define C64_LOAD_FRAGS(set, k32_)                                                                                                   \
    {                                                                                                                               \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                                          \
            const int rr_ = hf * 16 + (lane & 15);                                                                                  \
            const int sl_ = (((k32_) * 4 + (lane >> 4)) ^ ((rr_ >> 1) & 7)) * 16;                                                   \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                          \
                fa[set][i][hf] = *reinterpret_cast<const frag_t*>(sa + ((wr * TM + i) * 32 + rr_) * 128 + sl_);                     \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                          \
                fb[set][j][hf] = *reinterpret_cast<const frag_t*>(sb + ((wc * TN + j) * 32 + rr_) * 128 + sl_);                     \
        }                                                                                                                           \
    }
    // MFMAs of one 64-wide K slice; with ISSUE the LDS-DMA pieces of the next slice go out one at a time between groups of
    // MFMAs (their address arithmetic and issue slots sit in the matrix pipe's shadow instead of in front of it).
    constexpr int NMFMA = 2 * TM * TN * 4, NPIECE = A_PIECES + B_PIECES;
    constexpr int PSTEP = (NMFMA / C64_PSPREAD) / NPIECE > 0 ? (NMFMA / C64_PSPREAD) / NPIECE : 1;  // pieces go out in the first 1/C64_PSPREAD of the slice
    static_assert(PSTEP >= 1, "more staging pieces than MFMAs per K slice");
    // two fragment register sets: the second 32-wide step's fragments are read under the first step's MFMAs
    constexpr int FSETS = 2;
    // One fragment of the second 32-wide step, by its rank in the order the step's MFMAs need them: A0/0, B0/0, B0/1, A0/1, the other B
    // tiles, the other A tiles. With C64_FRAG_SPREAD these reads go out one at a time between the first step's MFMAs instead of
    // all 2 (TM + TN) of them in front of the slice's first MFMA, where all eight waves of the workgroup, released by the same
    // barrier, queue 192 KB of LDS reads at once.
    constexpr int NFRAG = 2 * (TM + TN);
    constexpr int FSTEP = (NMFMA / 2) / NFRAG > 0 ? (NMFMA / 2) / NFRAG : 1;
#define C64_LOAD_FRAG_RANK(set, k32_, rank_)                                                                                        \
    {                                                                                                                               \
        constexpr int rk_ = (rank_);                                                                                                \
        constexpr bool isa_ = rk_ == 0 || rk_ == 3 || rk_ >= 4 + 2 * (TN - 1);                                                      \
        constexpr int t_ = rk_ < 4 ? 0 : (isa_ ? 1 + (rk_ - 4 - 2 * (TN - 1)) / 2 : 1 + (rk_ - 4) / 2);                             \
        constexpr int hf_ = rk_ == 0 || rk_ == 1 ? 0 : (rk_ == 2 || rk_ == 3 ? 1 : (isa_ ? (rk_ - 4 - 2 * (TN - 1)) % 2 : (rk_ - 4) % 2)); \
        const int rr_ = hf_ * 16 + (lane & 15);                                                                                     \
        const int sl_ = (((k32_) * 4 + (lane >> 4)) ^ ((rr_ >> 1) & 7)) * 16;                                                       \
        if constexpr (isa_) fa[set][t_ < TM ? t_ : 0][hf_] = *reinterpret_cast<const frag_t*>(sa + ((wr * TM + t_) * 32 + rr_) * 128 + sl_); \
        else fb[set][t_ < TN ? t_ : 0][hf_] = *reinterpret_cast<const frag_t*>(sb + ((wc * TN + t_) * 32 + rr_) * 128 + sl_);        \
    }
#ifndef C64_FRAG_SPREAD
#define C64_FRAG_SPREAD 0
#endif
    constexpr bool SPREAD = C64_FRAG_SPREAD && BM * BN >= 256 * 256;  // (the 128-wide kernels pay for it with their third wave per SIMD: 166 -> 186 registers)
#define C64_KSLICE(ISSUE, nstage)                                                                                                   \
    {                                                                                                                               \
        frag_t fa[FSETS][TM][2], fb[FSETS][TN][2];                                                                                  \
        C64_LOAD_FRAGS(0, 0);                                                                                                       \
        _Pragma("unroll") for (int k32 = 0; k32 < 2; ++k32) {                                                                       \
            if (k32 < 1 && !SPREAD) C64_LOAD_FRAGS(1, 1);                                                                  \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                          \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                      \
                    _Pragma("unroll") for (int si = 0; si < 2; ++si)                                                                \
                        _Pragma("unroll") for (int sj = 0; sj < 2; ++sj) {                                                          \
                            acc[i][j][si][sj] = EPI == 2 ? Frag64<TI>::mfma16(fb[k32 % FSETS][j][sj], fa[k32 % FSETS][i][si], acc[i][j][si][sj]) \
                                                         : Frag64<TI>::mfma16(fa[k32 % FSETS][i][si], fb[k32 % FSETS][j][sj], acc[i][j][si][sj]); \
                            const int done_ = (((k32 * TM + i) * TN + j) * 2 + si) * 2 + sj + 1;                                    \
                            if (SPREAD && k32 == 0 && done_ % FSTEP == 0 && done_ / FSTEP <= NFRAG) {                      \
                                C64_SPREAD_ONE(done_ / FSTEP - 1);                                                                  \
                                __builtin_amdgcn_sched_barrier(0);                                                                  \
                            }                                                                                                       \
                            if (ISSUE) {                                                                                            \
                                if (done_ % PSTEP == 0 && done_ / PSTEP <= NPIECE) {                                                \
                                    C64_ISSUE_PIECE(nstage, done_ / PSTEP - 1);                                                     \
                                    __builtin_amdgcn_sched_barrier(0);                                                              \
                                }                                                                                                   \
                            }                                                                                                       \
                        }                                                                                                           \
        }                                                                                                                           \
    }
    // (the rank is a loop-derived value, constant after unrolling: dispatch it to the constexpr form)
#define C64_SPREAD_ONE(r_)                                                                                                          \
    {                                                                                                                               \
        switch (r_) {                                                                                                               \
            case 0: C64_LOAD_FRAG_RANK(1, 1, 0); break; case 1: C64_LOAD_FRAG_RANK(1, 1, 1); break;                                 \
            case 2: C64_LOAD_FRAG_RANK(1, 1, 2); break; case 3: C64_LOAD_FRAG_RANK(1, 1, 3); break;                                 \
            case 4: C64_LOAD_FRAG_RANK(1, 1, 4); break; case 5: C64_LOAD_FRAG_RANK(1, 1, 5); break;                                 \
            case 6: C64_LOAD_FRAG_RANK(1, 1, 6); break; case 7: C64_LOAD_FRAG_RANK(1, 1, 7); break;                                 \
            case 8: C64_LOAD_FRAG_RANK(1, 1, 8 < NFRAG ? 8 : 0); break; case 9: C64_LOAD_FRAG_RANK(1, 1, 9 < NFRAG ? 9 : 0); break; \
            case 10: C64_LOAD_FRAG_RANK(1, 1, 10 < NFRAG ? 10 : 0); break; case 11: C64_LOAD_FRAG_RANK(1, 1, 11 < NFRAG ? 11 : 0); break; \
            case 12: C64_LOAD_FRAG_RANK(1, 1, 12 < NFRAG ? 12 : 0); break; case 13: C64_LOAD_FRAG_RANK(1, 1, 13 < NFRAG ? 13 : 0); break; \
            case 14: C64_LOAD_FRAG_RANK(1, 1, 14 < NFRAG ? 14 : 0); break; case 15: C64_LOAD_FRAG_RANK(1, 1, 15 < NFRAG ? 15 : 0); break; \
            default: break;                                                                                                         \
        }                                                                                                                           \
    }
#define C64_FIRST_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    if constexpr (PH8) {
        // ---- 8-phase K loop (round 5; cdna_hip_programming.md section 5, "The 256^2 8-phase template", rebuilt here for the implicit-GEMM
        // gather). A K tile is four phases, one 64 x 32 quadrant of the wave's 128 x 64 outputs each (16 MFMAs), in the order
        // (qa, qb) = (0,0) (0,1) (1,1) (1,0); a phase reads only the fragments that are new to it and issues the staging units (two LDS-DMA
        // pieces per wave and unit) whose LDS rows have just been freed:
        //     phase 1: read A half 0 + B half 0 |  --                               phase 3: read A half 1 | issue B half 0 of tile t+2
        //     phase 2: read B half 1            | issue A half 0 of tile t+2        phase 4: (no read)     | issue B half 1, A half 1 of tile t+2
        // Waves 4-7 (wr == 1) run one barrier behind waves 0-3: while one wave of a SIMD is in its 16-MFMA cluster its partner reads
        // fragments and issues its pieces. The only wait on the vector-memory counter is phase 4's vmcnt(8): the four units of tile t+2
        // stay in flight across the tile boundary (never 0 inside the loop; every unit has at least four phases to land), everything
        // older -- all of tile t+1 -- has landed for this wave, and the barrier behind the wait publishes that to the others before
        // phase 1 of tile t+1 reads it (a read sits one phase AFTER the wait that retires the data). Write-after-read: every phase
        // retires its own fragment reads (lgkmcnt(0)) in front of its first barrier, so a unit may be restaged one phase after its
        // last read by either wave group (A half 0: read in phase 1, restaged in 2; A half 1: 3 -> 4; the B halves two phases later).
        // Past the last tile the units are dummies (tap 31: zero fill; weights re-read slice 0) so that the counts stay uniform. Per
        // accumulator the K order is the generic loop's: outputs are bit-identical to the TWO == 1 kernel (tests/test_conv_8phase.py).
        // Measured (scripts/exp_ph8.py, same process, random operands): FC1 1542 -> 1379 us, fpn_output2 1153 -> 1076 us; ablations
        // (scripts/exp_ph8_ablate.patch): without the pieces 1100 us, pieces that always hit L2 1247 us, without fragment reads 1333 us.
        const unsigned fo0 = (unsigned)((lane & 15) * 128 + ((((lane >> 4) ^ ((lane & 15) >> 1)) & 7) << 4));  // 32-wide K step 0; step 1: ^ 64
        frag_t fa[2][2][2], fb0[2][2], fb1[2][2];  // [row tile of the half][16-row sub-tile][32-wide K step]; [16-column sub-tile][K step]
#define P8_READ_A(qa_)                                                                                                              \
        {                                                                                                                           \
            _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                        \
                _Pragma("unroll") for (int si = 0; si < 2; ++si)                                                                    \
                    _Pragma("unroll") for (int k32 = 0; k32 < 2; ++k32)                                                             \
                        fa[ii][si][k32] = *reinterpret_cast<const frag_t*>(sa + ((wr * 4 + (qa_) * 2 + ii) * 32 + si * 16) * 128 + (fo0 ^ (unsigned)(k32 * 64))); \
        }
#define P8_READ_B(dst_, qb_)                                                                                                        \
        {                                                                                                                           \
            _Pragma("unroll") for (int sj = 0; sj < 2; ++sj)                                                                        \
                _Pragma("unroll") for (int k32 = 0; k32 < 2; ++k32)                                                                 \
                    dst_[sj][k32] = *reinterpret_cast<const frag_t*>(sb + ((wc * 2 + (qb_)) * 32 + sj * 16) * 128 + (fo0 ^ (unsigned)(k32 * 64))); \
        }
#define P8_MFMA4(qa_, fb_, qb_, k32_, ii_)                                                                                          \
        {                                                                                                                           \
            _Pragma("unroll") for (int si = 0; si < 2; ++si)                                                                        \
                _Pragma("unroll") for (int sj = 0; sj < 2; ++sj)                                                                    \
                    acc[(qa_) * 2 + (ii_)][qb_][si][sj] = Frag64<TI>::mfma16(fa[ii_][si][k32_], fb_[sj][k32_], acc[(qa_) * 2 + (ii_)][qb_][si][sj]); \
        }
        // one phase's 16 MFMAs; MID is placed behind the 12th of them (phase 1: the scalar K-state advance, in the matrix pipe's shadow).
        // Measured and dropped: the phase's pieces issued from inside the cluster (one / both of them: 3 % / 8 % slower), the waves of odd
        // wave columns issuing their pieces in front of their reads (4 % slower), reads retired behind the phase's first barrier (+-0).
#define P8_MFMA(qa_, fb_, qb_, st_, MID)                                                                                            \
        {                                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            __builtin_amdgcn_s_setprio(1);                                                                                          \
            P8_MFMA4(qa_, fb_, qb_, 0, 0);                                                                                          \
            P8_MFMA4(qa_, fb_, qb_, 0, 1);                                                                                          \
            P8_MFMA4(qa_, fb_, qb_, 1, 0);                                                                                          \
            MID;                                                                                                                    \
            P8_MFMA4(qa_, fb_, qb_, 1, 1);                                                                                          \
            __builtin_amdgcn_s_setprio(0);                                                                                          \
            P8_STAMP(st_);                                                                                                          \
            __builtin_amdgcn_s_barrier();                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }
#define P8_PIECE(stage_, q_)                                                                                                        \
        {                                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            C64_ISSUE_PIECE(stage_, (q_));                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }
#define P8_ISSUE(stage_, q0_)                                                                                                       \
        {                                                                                                                           \
            P8_PIECE(stage_, (q0_));                                                                                                \
            P8_PIECE(stage_, (q0_) + 1);                                                                                            \
        }
#define P8_ADV(tile_)                                                                                                               \
        {                                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            C64_ADVANCE();                                                                                                          \
            if ((tile_) >= nk) { tap = 31; kbyte = 0; }                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }
#define P8_UNIT_A0 0
#define P8_UNIT_A1 2
#define P8_UNIT_B0 (A_PIECES)
#define P8_UNIT_B1 (A_PIECES + 2)
        // prologue: tile 0 is in flight (C64_ISSUE(0) above, and the epilogue operands behind it); three units of tile 1 go out before the wait
        P8_ADV(1);
        P8_ISSUE(1, P8_UNIT_A0);
        P8_ISSUE(1, P8_UNIT_B0);
        P8_ISSUE(1, P8_UNIT_B1);
        P8_ISSUE(1, P8_UNIT_A1);
        c64_wait_vm_lgkm_barrier<8>();  // tile 0 (and every older load) has landed, for every wave
        C64_STAMP(1);
#ifdef C64_STAMPS
        unsigned long long p8s[24];
        for (int i = 0; i < 24; ++i) p8s[i] = 0;
#endif
        if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one barrier behind
        __builtin_amdgcn_sched_barrier(0);
        for (int ks = 0; ks < nk; ++ks) {
            const int cs = ks & 1;
            const unsigned char* sa = lds + cs * STAGE;
            const unsigned char* sb = sa + BM * 128;
            // phase 1 (the K state is one tile ahead: tile ks+1; it advances to tile ks+2 inside the cluster)
            P8_STAMP(0);
            P8_READ_A(0);
            P8_READ_B(fb0, 0);
            P8_STAMP(1);
            P8_STAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            P8_STAMP(3);
            P8_MFMA(0, fb0, 0, 4, P8_ADV(ks + 2));
            // phase 2
            P8_STAMP(5);
            P8_READ_B(fb1, 1);
            P8_STAMP(6);
            P8_ISSUE(cs, P8_UNIT_A0);  // tile ks+2: the rows phase 1 has just read (retired in front of its barrier)
            P8_STAMP(7);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            P8_STAMP(8);
            P8_MFMA(0, fb1, 1, 9, {});
            // phase 3
            P8_STAMP(10);
            P8_READ_A(1);
            P8_STAMP(11);
            P8_ISSUE(cs, P8_UNIT_B0);
            P8_STAMP(12);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            P8_STAMP(13);
            P8_MFMA(1, fb1, 1, 14, {});
            // phase 4
            P8_STAMP(15);
            P8_STAMP(16);
            P8_ISSUE(cs, P8_UNIT_B1);
            P8_ISSUE(cs, P8_UNIT_A1);  // (its rows were read in phase 3 and retired in front of that phase's barrier)
            P8_STAMP(17);
            c64_wait_vm_lgkm_barrier<8>();   // tile ks+1 complete; the four units of tile ks+2 stay in flight
            P8_STAMP(18);
            P8_MFMA(1, fb0, 0, 19, {});
            P8_STAMP(20);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();  // waves 0-3 take the barrier the others are one behind by
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy units have landed before LDS is reused
#ifdef C64_STAMPS
        if (a.p8 && (tid & 255) == 0)
            for (int i = 0; i < 24; ++i) a.p8[((long long)blockIdx.x * 2 + (tid >> 8)) * 24 + i] = p8s[i];
#endif
#undef P8_READ_A
#undef P8_READ_B
#undef P8_MFMA
#undef P8_MFMA4
#undef P8_PIECE
#undef P8_ADV
#undef P8_ISSUE
    } else if constexpr (TWO) {
        for (int ks = 0; ks < nk; ++ks) {
            if (ks == 0) { C64_FIRST_WAIT(); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();  // tile ks landed for every wave; every wave is done reading the other stage
            if (ks == 0) { C64_STAMP(1); }
            C64_ADVANCE();
            if (ks + 1 >= nk) { tap = 31; kbyte = 0; }  // nothing left to stage: tap 31 is never valid (zero fill), weights re-read slice 0
            const unsigned char* sa = lds + (ks & 1) * STAGE;
            const unsigned char* sb = sa + BM * 128;
            const int nstage = (ks + 1) & 1;
            C64_KSLICE(true, nstage);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last slice's dummy pieces have landed before LDS is reused
    } else {
        for (int ks = 0; ks < nk; ++ks) {
            if (ks == 0) { C64_FIRST_WAIT(); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();
            if (ks == 0) { C64_STAMP(1); }
            const unsigned char* sa = lds;
            const unsigned char* sb = sa + BM * 128;
            C64_KSLICE(false, 0);
            if (ks + 1 < nk) {
                __syncthreads();  // every wave has read the single staging buffer: refill it
                C64_ADVANCE();
                C64_ISSUE(0);
            }
        }
    }
#undef C64_KSLICE
#undef C64_SPREAD_ONE
#undef C64_LOAD_FRAG_RANK
#undef C64_LOAD_FRAGS
    __syncthreads();  // all waves done with the staging buffers before the epilogue reuses them
    C64_STAMP(2);

    if constexpr (EPI == 2) {
        // ---- chained 1x1 convolution (a bottleneck's conv2 -> conv3 + residual, osr_conv2d_chain_fwd). relu(conv + bias) of the tile is
        // parked in LDS in the storage dtype -- the same rounding point as the separate launches -- as two swizzled 64-channel slices
        // (the layout of a staged A tile). Each wave then keeps its 64 pixels x 128 channels as MFMA fragments in registers and walks
        // the chained layer's output channels 64 at a time: a stage = 64 weight rows x K = 128 (16 KB) through a three-slot LDS-DMA
        // ring, two stages ahead; the residual and bias of a stage are requested two stages ahead as well (buffer loads; a wave's
        // vector-memory operations return in order, so the counted waits below only ever wait for operations issued at least a
        // stage earlier). Weights are the A operand: a lane ends up with eight consecutive output channels of one pixel and stores
        // 16 bytes, no transposition. One asm statement per stage: s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier (the slot restaged behind
        // it was read in the previous stage: those reads have retired).
        static_assert(EPI != 2 || (BN == 128 && WN == 2 && TM == 2 && TWO == 1 && SPLIT == 0), "chained 1x1: 128 output channels, 64 x 64 per wave, two staging buffers");
        constexpr int CH_A = 0, CH_SL = BM * 128, CH_W = 2 * CH_SL, CH_STG = 16384, CH_NS = 8, CH_NR = 6, CH_PW = 16 / NW;  // 8 stages of 64 channels: cout3 == 512
        union FU { frag_t f; c64_u32x4 u; };
        const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int si = 0; si < 2; ++si) {
                    const int r = (wr * TM + i) * 32 + si * 16 + l15;
                    const int chunk = j * 4 + g;
                    frag_t t;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = acc[i][j][si][0][q] + cb2[j][q], v1 = acc[i][j][si][1][q] + cb2[j][4 + q];
                        t[q] = (TI)(p.relu ? fmaxf(v0, 0.f) : v0);
                        t[4 + q] = (TI)(p.relu ? fmaxf(v1, 0.f) : v1);
                    }
                    *reinterpret_cast<frag_t*>(lds + CH_A + wc * CH_SL + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = t;
                    if (a.mid_out) {  // (uniform) the same eight channels of pixel m0 + r, as the separate launch would have written them
                        const long long m = m0 + r;
                        if (m < e_M) *reinterpret_cast<frag_t*>(reinterpret_cast<TI*>(a.mid_out) + m * 128 + wc * 64 + chunk * 8) = t;
                    }
                }
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w3), 0, a.w3_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res), 0, a.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(e_out, 0, a.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias3), 0, (unsigned)a.cout3 * 4u, 0x00020000);
        unsigned w3_off[CH_PW];  // this lane's share of a stage: pieces wid * CH_PW + j of 16 (piece = K half (pc >> 3), rows (pc & 7) * 8 ..)
#pragma unroll
        for (int j = 0; j < CH_PW; ++j) {
            const int pc = wid * CH_PW + j, ksl = pc >> 3, row = (pc & 7) * 8 + lrow;
            const int chunk = slot ^ ((row >> 1) & 7);
            w3_off[j] = (unsigned)((c64_chain_perm(row) * 128 + ksl * 64 + chunk * 8) * 2);
        }
        unsigned o_off[TM][2];  // byte offset of this lane's pixel row in out / res (dense rows of cout3 elements), + its channel group
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const long long m = m0 + (wr * TM + i) * 32 + si * 16 + l15;
                o_off[i][si] = m < e_M ? (unsigned)((m * a.cout3 + wc * 32 + g * 8) * 2) : OOB_OFF;
            }
        const unsigned b3_off = (unsigned)((wc * 32 + g * 8) * 4);
        FU rr[3][TM][2];
        c64_u32x4 rb[3][2];
#define C64_CH_ISSUE(s_)                                                                                                             \
        {                                                                                                                            \
            _Pragma("unroll") for (int j_ = 0; j_ < CH_PW; ++j_) {                                                                   \
                const unsigned vo_ = w3_off[j_];                                                                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_void_t*)(lds + CH_W + ((s_) % 3) * CH_STG + (wid * CH_PW + j_) * 1024), 16, vo_, \
                                                         (s_) * 64 * 128 * 2, 0, 0);                                                  \
            }                                                                                                                        \
            __builtin_amdgcn_sched_barrier(0); /* the counted waits below rely on this issue order */                                \
        }
#define C64_CH_FETCH(s_)                                                                                                             \
        {                                                                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                                        \
                _Pragma("unroll") for (int si_ = 0; si_ < 2; ++si_) {                                                                \
                    rr[(s_) % 3][i_][si_].u = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o_off[i_][si_], (s_) * 128, 0);           \
                }                                                                                                                    \
            rb[(s_) % 3][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_b3, b3_off, (s_) * 256, 0);                                    \
            rb[(s_) % 3][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_b3, b3_off, (s_) * 256 + 16, 0);                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                       \
        }
        __syncthreads();  // the parked image is complete; the staging buffers of the K loop are free
        C64_CH_ISSUE(0);
        C64_CH_FETCH(0);
        C64_CH_ISSUE(1);
        C64_CH_FETCH(1);
        frag_t fp[TM][2][4];  // this wave's 64 pixels x 128 channels: B operand of every stage
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int si = 0; si < 2; ++si)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int rr_ = si * 16 + l15;
                    fp[i][si][k4] = *reinterpret_cast<const frag_t*>(lds + CH_A + (k4 >> 1) * CH_SL + ((wr * TM + i) * 32 + rr_) * 128 +
                                                                     ((((k4 & 1) * 4 + g) ^ ((rr_ >> 1) & 7)) << 4));
                }
        const int wrow_ = wc * 32 + l15;
        // One stage. The counted waits are exact: vector-memory operations issued behind the one waited for, per stage P = CH_PW
        // LDS-DMA pieces, R = CH_NR residual / bias loads, 4 stores -- nothing is issued for stages past the last one (a dummy load
        // whose result is unused would be removed by the compiler and the counts with it).
#define C64_CH_HAS(t_) ((t_) >= 0 && (t_) < CH_NS)
#define C64_CH_STAGE(S)                                                                                                              \
        {                                                                                                                            \
            constexpr int s = (S);                                                                                                   \
            constexpr int PR1 = C64_CH_HAS(s + 1) ? CH_PW + CH_NR : 0, PR2 = C64_CH_HAS(s + 2) ? CH_PW + CH_NR : 0;                  \
            constexpr int ST1 = s >= 1 ? 4 : 0, ST2 = s >= 2 ? 4 : 0;                                                                \
            /* stage s has landed for this wave: younger = R(s), stores(s-2), P(s+1), R(s+1), stores(s-1) */                         \
            c64_wait_vm_lgkm_barrier<CH_NR + ST2 + PR1 + ST1>();                                                                     \
            if constexpr (C64_CH_HAS(s + 2)) {                                                                                       \
                C64_CH_ISSUE(s + 2);                                                                                                 \
                C64_CH_FETCH(s + 2);                                                                                                 \
            }                                                                                                                        \
            const unsigned char* sb = lds + CH_W + (s % 3) * CH_STG;                                                                 \
            f32x4 a3[TM][2][2];                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                           \
                _Pragma("unroll") for (int si = 0; si < 2; ++si)                                                                     \
                    _Pragma("unroll") for (int sj = 0; sj < 2; ++sj) a3[i][si][sj] = f32x4{0.f, 0.f, 0.f, 0.f};                      \
            _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4) {                                                                       \
                frag_t fw[2];                                                                                                        \
                _Pragma("unroll") for (int sj = 0; sj < 2; ++sj) {                                                                   \
                    const int rw_ = wrow_ + sj * 16;                                                                                 \
                    fw[sj] = *reinterpret_cast<const frag_t*>(sb + (k4 >> 1) * 8192 + rw_ * 128 + ((((k4 & 1) * 4 + g) ^ ((rw_ >> 1) & 7)) << 4)); \
                }                                                                                                                    \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                       \
                    _Pragma("unroll") for (int si = 0; si < 2; ++si)                                                                 \
                        _Pragma("unroll") for (int sj = 0; sj < 2; ++sj)                                                             \
                            a3[i][si][sj] = Frag64<TI>::mfma16(fw[sj], fp[i][si][k4], a3[i][si][sj]);                                \
            }                                                                                                                        \
            /* the stage's residual and bias (requested two stages ago): younger = stores(s-2), P+R(s+1), stores(s-1), P+R(s+2) */   \
            c64_wait_vm<ST2 + PR1 + ST1 + PR2>();                                                                                    \
            union { c64_u32x4 u; float f[4]; } b0, b1;                                                                               \
            b0.u = rb[s % 3][0]; b1.u = rb[s % 3][1];                                                                                \
            FU o[TM][2];                                                                                                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                           \
                _Pragma("unroll") for (int si = 0; si < 2; ++si) {                                                                   \
                    const frag_t rv = rr[s % 3][i][si].f;                                                                            \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                  \
                        float v = (e < 4 ? a3[i][si][0][e & 3] : a3[i][si][1][e & 3]) + (e < 4 ? b0.f[e & 3] : b1.f[e & 3]);         \
                        v += (float)rv[e];                                                                                           \
                        o[i][si].f[e] = (TI)fmaxf(v, 0.f);                                                                           \
                    }                                                                                                                \
                }                                                                                                                    \
            /* The four stores go out together at the end of the stage. A 16-byte buffer store reads its data after it has issued;   \
               with a REGISTER in the soffset field LLVM's hazard rule inserts no wait state and the next instruction may overwrite   \
               the data (round 3, measured: under load from a second stream lanes 12-15 of every row of 16 stored the NEXT value of   \
               the first data register). The stage's channel offset therefore rides in the instruction's immediate offset field      \
               (s * 128 <= 896 < 4096, folded from the voffset sum) and soffset is the constant 0: the compiler's own wait state      \
               applies -- the hazard class is gone, not padded (tests/test_conv_chain.py keeps the two-stream scenario as a gate). */  \
            __builtin_amdgcn_sched_barrier(0);                                                                                       \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                           \
                _Pragma("unroll") for (int si = 0; si < 2; ++si) /* (a row beyond M: offset outside the buffer, store dropped) */    \
                    __builtin_amdgcn_raw_buffer_store_b128(o[i][si].u, rs_out, o_off[i][si] + s * 128, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                       \
        }
        static_assert(CH_NS == 8, "eight stages written out");
        C64_CH_STAGE(0) C64_CH_STAGE(1) C64_CH_STAGE(2) C64_CH_STAGE(3) C64_CH_STAGE(4) C64_CH_STAGE(5) C64_CH_STAGE(6) C64_CH_STAGE(7)
#undef C64_CH_STAGE
#undef C64_CH_HAS
#undef C64_CH_ISSUE
#undef C64_CH_FETCH
        C64_STAMP(3);
        return;
    }

    if constexpr (EPI == 1) {
        // ---- fused CF-RPN tail (classification_free_rpn.py:159-161): t = relu(conv + bias) is parked in LDS in the
        // storage dtype (exactly what the unfused path writes to HBM); ||t||^2 and the five 1x1 dot products come from 32 MFMAs per
        // wave on that tile (below), then one lane per pixel normalises, adds the 1x1 biases and applies the sigmoid. ----
        constexpr int LDT = 256 + 8;  // elements per t row (528 B: 16-B aligned, conflict-free chunk walk)
        TI* s_t = reinterpret_cast<TI*>(lds);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wr * TM + i) * 32 + C64_ROW(r);
                    const int col = (wc * TN + j) * 32 + C64_COL(r);
                    s_t[row * LDT + col] = (TI)fmaxf(C64_ACC(i, j, r) + tbias[j][(r >> 2) & 1], 0.f);
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // a few accumulators at a time: keeps the VGPR budget of the K loop
                }
        __syncthreads();
        if (e_out) {  // the training step keeps the hidden state for the head's backward: the tile's rows, 16 bytes per thread and store
            TI* __restrict__ tout = reinterpret_cast<TI*>(e_out);
            for (int q = tid; q < BM * 32; q += NT) {
                const int row = q >> 5, ch8 = (q & 31) * 8;
                const long long m = m0 + row;
                if (m < e_M) *reinterpret_cast<frag_t*>(tout + m * 256 + ch8) = *reinterpret_cast<const frag_t*>(s_t + row * LDT + ch8);
            }
        }
        // The five 1x1 dot products and ||t||^2 on the matrix cores (round 5; until then two threads per pixel, 768 fp32 FMAs + 170 LDS reads
        // each: 0.2 ms of the merged head launch). Per wave, its 32 pixels x 256 channels of s_t are the A operand (16 fragments);
        //   D  = T . W16^T  (16 MFMAs): columns q, 5 + q, 10 + q of a pixel's row = its dot products with the hi / mid / lo terms of tail row q
        //                    (products of two storage-dtype values are exact in fp32; the sum of the three columns is the fp32-weight dot
        //                    product up to the accumulation order);
        //   D2 = T . T^T    (16 MFMAs, B operand = the A fragments themselves): the diagonal is ||t||^2.
        // The accumulators go through a (rows, 16) fp32 slab in LDS -- it takes the place of the weight matrix, once every wave holds its B
        // fragments --, then one lane per pixel normalises, adds the biases and applies the sigmoid.
        const TI* s_w16 = reinterpret_cast<const TI*>(lds + a.tail_lds_off);
        float* s_d = reinterpret_cast<float*>(lds + a.tail_lds_off);
        frag_t wb[8];
#pragma unroll
        for (int ks8 = 0; ks8 < 8; ++ks8) wb[ks8] = *reinterpret_cast<const frag_t*>(s_w16 + (lane & 15) * C64_TW_LD + ks8 * 32 + (lane >> 4) * 8);
        __syncthreads();  // every wave has its weight fragments: the slab may overwrite the matrix
        f32x4 dacc[2], sacc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) { dacc[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; sacc[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks8 = 0; ks8 < 8; ++ks8)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const frag_t ta = *reinterpret_cast<const frag_t*>(s_t + (wid * 32 + rt * 16 + (lane & 15)) * LDT + ks8 * 32 + (lane >> 4) * 8);
                dacc[rt] = Frag64<TI>::mfma16(ta, wb[ks8], dacc[rt]);
                sacc[rt] = Frag64<TI>::mfma16(ta, ta, sacc[rt]);
            }
        // D element q of a lane: row (lane >> 4) * 4 + q, column lane & 15. Column 15 of D is zero (zero weight row): the diagonal of D2
        // goes there (row r's lane is the one with column r, i.e. (lane & 15) >> 2 == lane >> 4, element (lane & 3)); same-wave LDS writes land in order
        const int dcol = lane & 15, dgrp = lane >> 4;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float* sd = s_d + (wid * 32 + rt * 16) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) sd[(dgrp * 4 + q) * 16 + dcol] = dacc[rt][q];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float* sd = s_d + (wid * 32 + rt * 16) * 16;
            const float dg = (dcol & 3) == 0 ? sacc[rt][0] : (dcol & 3) == 1 ? sacc[rt][1] : (dcol & 3) == 2 ? sacc[rt][2] : sacc[rt][3];
            if ((dcol >> 2) == dgrp) sd[dcol * 16 + 15] = dg;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's slab rows are written
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) {
            const int row = wid * 32 + lane;
            const long long m = m0 + row;
            const float4 v0 = *reinterpret_cast<const float4*>(s_d + row * 16), v1 = *reinterpret_cast<const float4*>(s_d + row * 16 + 4);
            const float4 v2 = *reinterpret_cast<const float4*>(s_d + row * 16 + 8), v3 = *reinterpret_cast<const float4*>(s_d + row * 16 + 12);
            if (m < e_M) {
                const float inv = 1.0f / fmaxf(sqrtf(v3.w), 1e-12f);
                const float d0 = v0.x + (v1.y + v2.z), d1 = v0.y + (v1.z + v2.w), d2 = v0.z + (v1.w + v3.x), d3 = v0.w + (v2.x + v3.y), d4 = v1.x + (v2.y + v3.z);
                // (undo each tail row's power-of-two weight scale: exact)
                const unsigned int* s_tmax = reinterpret_cast<const unsigned int*>(lds + a.tail_lds_off + C64_TAIL_LDS(BM));
#define C64_TAIL_UNSCALE(q) (inv * __uint_as_float((C64_TAIL_EXP(s_tmax[q]) - 10u) << 23))
                *reinterpret_cast<float4*>(e_tail_deltas + m * 4) = make_float4(d0 * C64_TAIL_UNSCALE(0) + a.tail_b[0], d1 * C64_TAIL_UNSCALE(1) + a.tail_b[1],
                                                                                d2 * C64_TAIL_UNSCALE(2) + a.tail_b[2], d3 * C64_TAIL_UNSCALE(3) + a.tail_b[3]);
                e_tail_ctr[m] = 1.0f / (1.0f + expf(-(d4 * C64_TAIL_UNSCALE(4) + a.tail_b[4])));
#undef C64_TAIL_UNSCALE
            }
        }
        return;
    }

    // ---- epilogue: wave-private fp32 slab (32 rows x 64 columns, one pair of N tiles at a time) -> 8 channels per lane:
    //      bias + residual / FPN 2x upsample-add + ReLU + convert on 16-byte row segments, coalesced along channels ----
    float* slab = reinterpret_cast<float*>(lds) + wid * 32 * EPI_LD;
    TO* __restrict__ out = reinterpret_cast<TO*>(e_out) + (SPLIT ? (long long)ksp * a.split_stride : 0ll);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jp = 0; jp < TNP; ++jp) {
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    slab[C64_ROW(r) * EPI_LD + j2 * 32 + C64_COL(r)] = C64_ACC(i, jp * 2 + j2, r);
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int co = n0 + (wc * TN + jp * 2) * 32 + cseg;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int row = pass * RPP + (lane >> 3);
                const long long m = m0 + (wr * TM + i) * 32 + row;
                if (m < e_M && co < e_cout) {
                    const float4 v0 = *reinterpret_cast<const float4*>(slab + row * EPI_LD + cseg);
                    const float4 v1 = *reinterpret_cast<const float4*>(slab + row * EPI_LD + cseg + 4);
                    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bias8[jp][e];
                    // element offsets of this lane's eight channels in the output (= the ReLU mask's layout) and in the residual. Dense rows
                    // everywhere (pw_dense, wave-uniform): ONE 32-bit offset m * cout + co serves all three; else row -> (image, y, x) and strides
                    long long o_out, o_res = 0;
                    const bool need_res = !PRE_RES && p.res_mode != 0;  // (the 128-wide tiles hold the residual in registers already)
                    if ((a.pw_dense & 2) && (!need_res || (a.pw_dense & 4))) {
                        o_out = o_res = (long long)((unsigned)m * (unsigned)e_cout + (unsigned)co);
                    } else {
                        C64_ROW_TO_NHW(m, nimg, oh, ow);
                        o_out = (long long)nimg * e_osn + (long long)oh * e_osh + (long long)ow * e_osw + co;
                        if (need_res) {
                            const int rh = p.res_mode == 2 ? (oh >> 1) : oh, rw = p.res_mode == 2 ? (ow >> 1) : ow;
                            o_res = (long long)nimg * p.res_stride_n + (long long)rh * p.res_stride_h + (long long)rw * p.res_stride_w + co;
                        }
                    }
                    if (p.res_mode != 0) {
                        frag_t rv;
                        if constexpr (PRE_RES) {
                            rv = rres[i][pass];
                        } else {
                            rv = *reinterpret_cast<const frag_t*>(res + o_res);
                        }
                        if (p.res_mode == 3) {  // backward of a ReLU: keep the gradient where the forward activation was positive
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (float)rv[e] > 0.f ? v[e] : 0.f;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                        }
                    }
                    if (e_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (a.mask) {
                        const frag_t mv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const TI*>(a.mask) + o_out);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)mv[e] > 0.f ? v[e] : 0.f;
                    }
                    store8_64<TO>(out + o_out, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    C64_STAMP(3);
}

static size_t conv64_lds_bytes(int bm, int bn, int two_stage, int nw = 4) {
    const size_t stage = (size_t)(bm + bn) * 128, epi = (size_t)nw * 32 * 68 * 4;
    const size_t stages = two_stage ? 2 * stage : stage;
    return stages > epi ? stages : epi;
}

// Tuning constants. A -DOSR_EXPERIMENT build (never shipped; scripts/ab_*.sh) reads them from the environment instead, so
// that one box can compare settings; the product library has no environment dependence.
#ifdef OSR_EXPERIMENT
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define OSR_KNOB(name, dflt) ([] { static const int v = env_int(name, dflt); return v; }())
#else
#define OSR_KNOB(name, dflt) (dflt)
#endif
#ifdef OSR_EXPERIMENT
static int ph8_enabled() { return env_int("OSR_CONV_PH8", 1); }  // (read at every launch: one process can compare the two K loops, scripts/exp_ph8.py)
#endif
static int rpn_big_min_tiles() { return OSR_KNOB("OSR_RPN_BIG_MIN_TILES", 512); }  // fused CF-RPN head: 256-row tiles from this many tiles on
static int tap_minor_default() { return OSR_KNOB("OSR_CONV_TAP_MINOR", 1); }

template <class K>
static void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Tile configurations of the plain conv / FC kernel.
enum Conv64Tile { T128x128_1 = 1, T128x128_2, T256x256_2, T128x256_1, T256x128_1, T128x64_1, T128x64_2 };
static int force_tile() { return OSR_KNOB("OSR_CONV_FORCE_TILE", 0); }  // diagnostic: one configuration for every layer it fits

template <class TI, class TO, int BM, int BN, int WM, int WN, int TWO, int SPLIT = 0>
static void conv64_launch_tile(Conv64Args& a, hipStream_t st) {
    constexpr int NW = WM * WN;
    a.two_stage = TWO;
    a.tiles_m = (int)((a.M + BM - 1) / BM);
    a.tiles_n = (a.p.cout + BN - 1) / BN;
    if (a.ntile <= 0) { a.tile0 = 0; a.ntile = a.tiles_m * a.tiles_n; a.ksplit = 1; }  // the whole grid in one launch
    const size_t lds = conv64_lds_bytes(BM, BN, TWO, NW);
    if (lds > 64 * 1024) {
        static osr_dev_mask attr{0};  // (one per instantiation of this function template)
        osr_once_per_device(attr, [] { allow_big_lds(conv_igemm64_kernel<TI, TO, BM, BN, WM, WN, 0, TWO, SPLIT>); });
    }
    hipLaunchKernelGGL((conv_igemm64_kernel<TI, TO, BM, BN, WM, WN, 0, TWO, SPLIT>), dim3((unsigned)a.ntile * (SPLIT ? a.ksplit : 1)), dim3(NW * 64), lds, st, a);
}

// Tile choice by a small cost model instead of per-layer thresholds. Measured on MI355X (scripts/ab_tiles.sh): with the
// staging pieces interleaved into the MFMA stream every configuration saturates at the same ~4.45 TFLOP/s per CU once a CU
// holds its full complement of workgroups, so what separates them is (i) how many dispatch rounds the grid needs at
// `occ` workgroups per CU (VGPR / LDS limited) and how empty the last round is, and (ii) whether a K slice is bound by
// the matrix pipe (r residents x w) or by the L2 -> LDS latency (single buffer: L + w, double buffer: max(L, r x w)).
//   time = full_rounds x tile(occ) + tile(residents of the last partial round),  tile(r) = fixed + nk x slice(r)
// The 256-wide tiles read the residual in the epilogue (no prefetch under the K loop), which measured 1.6x slower on the
// HBM-bound residual layers: they are not offered to a residual layer with fewer than 18 K slices.
#ifndef C64_BIG_CAP
#define C64_BIG_CAP 5.2e6  // per-CU rate of the 256 x 256 tile under load, FLOP per us
#endif
struct TileCfg { int id, bm, bn, two, occ, pre_res; double cap; };  // cap: per-CU MFMA rate under load, FLOP per us
static const TileCfg kTileCfgs[] = {
    {T128x128_1, 128, 128, 0, 3, 1, 4.45e6}, {T128x128_2, 128, 128, 1, 2, 1, 4.45e6}, {T256x256_2, 256, 256, 1, 1, 0, C64_BIG_CAP},
    {T128x256_1, 128, 256, 0, 2, 0, 4.45e6}, {T256x128_1, 256, 128, 0, 2, 0, 4.45e6}, {T128x64_1, 128, 64, 0, 4, 1, 4.45e6},
    {T128x64_2, 128, 64, 1, 3, 1, 4.45e6},
};

static double conv64_tile_us(const TileCfg& c, int nk, int residents, bool res) {
    const double kLat = 1.0;  // L2 -> LDS latency of one staged slice (us)
    const double w = (double)c.bm * c.bn * 128.0 / c.cap;  // (the 256 x 256 tile measured 1.61 us per slice on fpn_output2: 5.2 TFLOP/s per CU)
    const double busy = residents * w;
    const double slice = c.two ? (busy > kLat + 0.1 ? busy : kLat + 0.1) : (busy > kLat + w ? busy : kLat + w);
    const double fixed = 1.5 + (c.two ? kLat : 0.0) + (double)c.bm * c.bn / 16384.0 * 1.3 * (res ? 1.5 : 1.0);
    return fixed + nk * slice;
}

// When the caller runs several streams side by side (osr_conv_params.concurrency >= 2: the engine's micro-batch streams) the
// modelled time of the 256 x 256 tile is scaled by 0.8 (OSR_CONV_MODEL_BIG, percent): the empty part of a
// one-workgroup-per-CU round is then filled by the other stream's launches, and the big tile
// moves half the L2 -> LDS and LDS -> register bytes per FLOP of the 128 x 128 tile, which is what counts once both streams
// compete for a CU (same-box end-to-end A/B with the per-configuration rates above: 1.00 -> 1169, 0.85 -> 1221, 0.75 -> 1219,
// 0.65 -> 1208 img/s).
static double model_big_scale() { return OSR_KNOB("OSR_CONV_MODEL_BIG", 80) / 100.0; }

static int conv64_pick_tile(const Conv64Args& a) {
    const int nk = a.K / 64, cout = a.p.cout;
    const bool res = a.p.res_mode != 0;
    const int f = force_tile();
    if (f != 0) {
        const bool fits = (f == T256x256_2 || f == T128x256_1) ? cout % 256 == 0 : f == T256x128_1 ? cout % 128 == 0 : true;
        if (fits && !((f == T128x128_2 || f == T128x64_2 || f == T256x256_2) && nk < 2)) return f;
    }
    int best = T128x128_1;
    double best_us = 1e30;
    for (const TileCfg& c : kTileCfgs) {
        if (c.bn == 256 && cout % 256 != 0) continue;
        if (cout <= 64 && c.bn != 64) continue;          // narrow layers: no half-empty N tiles
        // the 256-wide tiles read the residual in the epilogue, unprefetched: not for the short-K (HBM-bound) residual layers. From 18 K slices on
        // (a 3 x 3 layer of >= 128 channels: the data gradients of fpn_output2/3 with their ReLU-mask / gradient-sum epilogues) the epilogue is
        // a few per cent of the tile and the 256 x 256 8-phase loop wins (round 5: fpn_output2's dgrad 1.65 -> 1.2 ms)
        if (res && !c.pre_res && nk < 18) continue;
        if (c.two && nk < 2) continue;
        const long long tiles = (a.M + c.bm - 1) / c.bm * ((cout + c.bn - 1) / c.bn), slots = 256ll * c.occ;
        const long long full = tiles / slots, rem = tiles % slots;
        double us = (double)full * conv64_tile_us(c, nk, c.occ, res);
        if (rem) us += conv64_tile_us(c, nk, (int)((rem + 255) / 256), res);
        if (c.id == T256x256_2 && a.p.concurrency >= 2) us *= model_big_scale();
        if (us < best_us) { best_us = us; best = c.id; }
    }
    return best;
}

// ---- split-K of the tail round ---------------------------------------------------------------------------------------
// A grid of T tiles on S = 256 x occ slots runs floor(T / S) full dispatch rounds and one round that is only (T mod S) / S full.
// For a deep-K 1x1 / FC layer (FC1: 1072 tiles of 256 x 256 on 256 slots, 196 K slices: 4.19 rounds, the fifth one 19 % full)
// the partial round is cut along K instead: its tiles go to a second launch in which ksplit workgroups share each tile's K loop
// and write fp32 partial sums, and a small third launch adds them up and applies the epilogue. The partial sums are summed in
// a fixed order: the result does not depend on timing.
struct SplitPlan { int tile_id, tail_mtiles, ksplit; long long m_tail0, ws_bytes; };

static bool conv64_plan_split(const Conv64Args& a, SplitPlan* sp) {
    const osr_conv_params& p = a.p;
    if (a.stem || p.res_mode != 0 || a.mask || p.stride_h != 1 || p.stride_w != 1 || p.cin % 64 != 0) return false;
    if (p.out_stride_w != p.cout || p.out_stride_h != (long long)p.wo * p.cout || p.out_stride_n != (long long)p.ho * p.wo * p.cout) return false;  // dense output rows
    const int nk = a.K / 64;
    // the partial sums (tail rows x cout x 4 B x ksplit, written and read once) must be small beside the K loop: 48 slices for a 1 x 1 /
    // FC layer, 32 for a KH x KW convolution (round 4: the 3 x 3 layers of res4 / FPN p4, 1050 tiles of 128 x 128 on 768 slots)
    if (nk < (p.kh * p.kw > 1 ? 32 : 48)) return false;
    const int id = conv64_pick_tile(a);
    if (id != T256x256_2 && id != T128x128_1 && id != T128x256_1) return false;  // the tile shapes with a split-K tail instantiation
    const TileCfg* c = nullptr;
    for (const TileCfg& k : kTileCfgs) if (k.id == id) c = &k;
    if (!c) return false;
    const long long tiles_m = (a.M + c->bm - 1) / c->bm, tiles_n = (p.cout + c->bn - 1) / c->bn, tiles = tiles_m * tiles_n, slots = 256ll * c->occ;
    const long long full = tiles / slots, rem = tiles % slots;
    if (full < 1 || rem == 0 || rem * 2 > slots) return false;
    // a K x K convolution needs at least two full rounds in front of the tail: with one (res4's 3 x 3 layers at batch 16, 1050 tiles on
    // 768 slots) the two extra launches and the partial sums cost more than the half-empty round (measured 100 -> 115 us), with four
    // (fpn_output3 on 256 x 256 tiles) the split takes 380 -> 321 us (scripts/exp_split3x3.py)
    if (p.kh * p.kw > 1 && full < 2) return false;
    const long long tail_mt = (rem + tiles_n - 1) / tiles_n;  // whole rows of M tiles
    long long ks = slots / (tail_mt * tiles_n);
    if (ks > 8) ks = 8;
    if (ks > nk / 8) ks = nk / 8;
    if (ks < 2) return false;
    sp->tile_id = id; sp->tail_mtiles = (int)tail_mt; sp->ksplit = (int)ks;
    sp->m_tail0 = (tiles_m - tail_mt) * c->bm;
    sp->ws_bytes = ks * (a.M - sp->m_tail0) * p.cout * 4;
    return true;
}

template <class TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, long long split_stride, long long n4, int cout,
                                                            const float* __restrict__ bias, int relu, TO* __restrict__ out) {
    // out[i] = act(sum_s ws[s][i] + bias[i % cout]), four elements per thread (cout % 8 == 0)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(ws + i * 4);
        for (int s = 1; s < ksplit; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(ws + s * split_stride + i * 4);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const float4 b = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % cout));
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        out[i * 4 + 0] = osr_from_float<TO>(v.x); out[i * 4 + 1] = osr_from_float<TO>(v.y);
        out[i * 4 + 2] = osr_from_float<TO>(v.z); out[i * 4 + 3] = osr_from_float<TO>(v.w);
    }
}

template <class TI, class TO> static void conv64_dispatch_tile(int id, Conv64Args& a, hipStream_t st);

template <class TI, class TO>
static osr_status conv64_launch(Conv64Args& a, hipStream_t st) {
    a.tail_lds_off = 0;
    a.tile0 = 0; a.ntile = 0; a.ksplit = 1; a.split_stride = 0;
    SplitPlan sp;
    if (a.p.workspace && conv64_plan_split(a, &sp) && a.p.workspace_bytes >= sp.ws_bytes && (((uintptr_t)a.p.workspace) & 15) == 0) {
        const TileCfg* c = nullptr;
        for (const TileCfg& k : kTileCfgs) if (k.id == sp.tile_id) c = &k;
        const int tiles_m = (int)((a.M + c->bm - 1) / c->bm), tiles_n = (a.p.cout + c->bn - 1) / c->bn;
        // 1. the full rounds
        Conv64Args m = a;
        m.tile0 = 0; m.ntile = (tiles_m - sp.tail_mtiles) * tiles_n; m.ksplit = 1;
        conv64_dispatch_tile<TI, TO>(sp.tile_id, m, st);
        // 2. the tail tiles, ksplit workgroups each, raw fp32 partial sums into the workspace slabs (rows relative to m_tail0)
        const long long tail_rows = a.M - sp.m_tail0;
        Conv64Args t = a;
        t.tile0 = m.ntile; t.ntile = sp.tail_mtiles * tiles_n; t.ksplit = sp.ksplit;
        t.split_stride = tail_rows * a.p.cout;
        t.out = reinterpret_cast<float*>(a.p.workspace) - sp.m_tail0 * a.p.cout;  // the epilogue adds row * cout: slab row 0 = output row m_tail0
        t.p.relu = 0; t.p.out_dtype = OSR_F32;
        switch (sp.tile_id) {
            case T256x256_2: conv64_launch_tile<TI, float, 256, 256, 2, 4, 2, 1>(t, st); break;
            case T128x256_1: conv64_launch_tile<TI, float, 128, 256, 2, 2, 0, 1>(t, st); break;
            default: conv64_launch_tile<TI, float, 128, 128, 2, 2, 0, 1>(t, st); break;
        }
        // 3. fixed-order sum of the slabs + bias + ReLU + conversion
        const long long n4 = tail_rows * a.p.cout / 4;
        long long blocks = (n4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel<TO>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float*>(a.p.workspace), sp.ksplit,
                           t.split_stride, n4, a.p.cout, a.bias, a.p.relu, reinterpret_cast<TO*>(a.out) + sp.m_tail0 * a.p.cout);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { osr_set_error("osr_conv2d_fwd(bk64, split-K tail): launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
        return OSR_OK;
    }
    conv64_dispatch_tile<TI, TO>(conv64_pick_tile(a), a, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_conv2d_fwd(bk64): launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

template <class TI, class TO>
static void conv64_dispatch_tile(int id, Conv64Args& a, hipStream_t st) {
    switch (id) {
        case T128x128_1: conv64_launch_tile<TI, TO, 128, 128, 2, 2, 0>(a, st); break;
        case T128x128_2: conv64_launch_tile<TI, TO, 128, 128, 2, 2, 1>(a, st); break;
#ifdef C64_BIG_W4
        case T256x256_2: conv64_launch_tile<TI, TO, 256, 256, 2, 2, 1>(a, st); break;
#else
        case T256x256_2:
#ifdef OSR_EXPERIMENT
            if (!ph8_enabled()) { conv64_launch_tile<TI, TO, 256, 256, 2, 4, 1>(a, st); break; }  // the round-4 K loop, for same-process comparisons
#endif
            conv64_launch_tile<TI, TO, 256, 256, 2, 4, 2>(a, st);
            break;
#endif
        case T128x256_1: conv64_launch_tile<TI, TO, 128, 256, 2, 2, 0>(a, st); break;
        case T256x128_1: conv64_launch_tile<TI, TO, 256, 128, 2, 2, 0>(a, st); break;
        case T128x64_1: conv64_launch_tile<TI, TO, 128, 64, 4, 1, 0>(a, st); break;
        default: conv64_launch_tile<TI, TO, 128, 64, 4, 1, 1>(a, st); break;
    }
}

// Host-side description of how osr_conv2d_fwd covers this layer (tests).
int osr_conv64_describe(const osr_conv_params* p, int has_workspace, char* buf, int n) {
    Conv64Args a;
    a.p = *p; a.mask = nullptr;
    a.M = (long long)p->n * p->ho * p->wo;
    a.K = p->kh * p->kw * p->cin;
    a.stem = (p->pad_mode == 1 && p->cin == 32) ? 1 : 0;
    auto nm = [](int id) { for (const TileCfg& k : kTileCfgs) if (k.id == id) return k; return kTileCfgs[0]; };
    SplitPlan sp;
    if (has_workspace && conv64_plan_split(a, &sp)) {
        const TileCfg c = nm(sp.tile_id);
        return snprintf(buf, n, "%dx%d/%d rows [0,%lld) + split-K x%d tail rows [%lld,%lld) + reduce", c.bm, c.bn, c.two + 1, sp.m_tail0, sp.ksplit, sp.m_tail0, a.M);
    }
    const TileCfg c1 = nm(conv64_pick_tile(a));
    return snprintf(buf, n, "%dx%d/%d rows [0,%lld)", c1.bm, c1.bn, c1.two + 1, a.M);
}

// Bytes of workspace with which osr_conv2d_fwd cuts this layer's partial last dispatch round along K (0: not applicable).
long long osr_conv64_split_workspace_bytes(const osr_conv_params* p) {
    Conv64Args a;
    a.p = *p; a.mask = nullptr;
    a.M = (long long)p->n * p->ho * p->wo;
    a.K = p->kh * p->kw * p->cin;
    a.stem = (p->pad_mode == 1 && p->cin == 32) ? 1 : 0;
    SplitPlan sp;
    if (p->cin % 64 != 0 || a.M >= (1ll << 31) - 1024) return 0;
    return conv64_plan_split(a, &sp) ? sp.ws_bytes : 0;
}

template <class TI>
static osr_status cfrpn_fused_launch(Conv64Args& a, hipStream_t st) {
    a.tiles_n = 1;
    a.tile0 = 0; a.ksplit = 1; a.split_stride = 0;
    if ((a.M + 255) / 256 >= rpn_big_min_tiles() && a.K / 64 >= 8) {
        a.two_stage = 1;
        a.tiles_m = (int)((a.M + 255) / 256);
        a.ntile = a.tiles_m;
        const size_t t_bytes = (size_t)256 * (256 + 8) * 2, stages = (size_t)2 * (256 + 256) * 128;
        a.tail_lds_off = (int)(t_bytes > stages ? t_bytes : stages);
        const size_t lds = (size_t)a.tail_lds_off + C64_TAIL_LDS(256) + 32;
        static osr_dev_mask attr8{0};
        osr_once_per_device(attr8, [] {
            allow_big_lds(conv_igemm64_kernel<TI, TI, 256, 256, 2, 4, 1, 2>);
#ifdef OSR_EXPERIMENT
            allow_big_lds(conv_igemm64_kernel<TI, TI, 256, 256, 2, 4, 1, 1>);
#endif
        });
#ifdef OSR_EXPERIMENT
        if (!ph8_enabled()) hipLaunchKernelGGL((conv_igemm64_kernel<TI, TI, 256, 256, 2, 4, 1, 1>), dim3((unsigned)a.tiles_m), dim3(512), lds, st, a);
        else
#endif
        hipLaunchKernelGGL((conv_igemm64_kernel<TI, TI, 256, 256, 2, 4, 1, 2>), dim3((unsigned)a.tiles_m), dim3(512), lds, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { osr_set_error("osr_cfrpn_head_fwd: launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
        return OSR_OK;
    }
    a.two_stage = 0;
    a.tiles_m = (int)((a.M + 127) / 128);
    a.ntile = a.tiles_m;
    const size_t t_bytes = (size_t)128 * (256 + 8) * 2, stage = (size_t)(128 + 256) * 128;
    a.tail_lds_off = (int)(t_bytes > stage ? t_bytes : stage);
    const size_t lds = (size_t)a.tail_lds_off + C64_TAIL_LDS(128) + 32;
    static osr_dev_mask attr{0};
    osr_once_per_device(attr, [] { allow_big_lds(conv_igemm64_kernel<TI, TI, 128, 256, 2, 2, 1, 0>); });
    hipLaunchKernelGGL((conv_igemm64_kernel<TI, TI, 128, 256, 2, 2, 1, 0>), dim3((unsigned)a.tiles_m), dim3(256), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_cfrpn_head_fwd: launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

// Returns 1 when this fast path can take the problem (all arguments already validated by osr_conv2d_fwd).
int osr_conv64_eligible(const osr_conv_params* p, long long in_bytes, long long w_bytes) {
    const bool stem = p->pad_mode == 1 && p->cin == 32 && p->kw == 1 && (p->kh % 2) == 0;
    if (!stem && p->cin % 64 != 0) return 0;
    if (!stem && p->kh * p->kw > 31) return 0;  // per-row tap validity mask: 31 bits
    if ((long long)p->n * p->ho * p->wo >= (1ll << 31) - 1024) return 0;  // 32-bit row indices (fastdiv)
    if (in_bytes <= 0 || w_bytes <= 0 || in_bytes >= (1ll << 31) - 4096 || w_bytes >= (1ll << 31) - 4096) return 0;
    return 1;
}

osr_status osr_conv64_run(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual, const void* mask,
                          void* out, long long in_bytes, long long w_bytes, hipStream_t st) {
    Conv64Args a;
    a.p = *p; a.in = in; a.w = weight; a.bias = bias; a.res = residual; a.mask = mask; a.out = out;
    a.M = (long long)p->n * p->ho * p->wo;
    a.K = p->kh * p->kw * p->cin;
    a.div_howo = fastdiv_make((unsigned)(p->ho * p->wo));
    a.div_wo = fastdiv_make((unsigned)p->wo);
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
    a.stem = (p->pad_mode == 1 && p->cin == 32) ? 1 : 0;
    a.tap_minor = (!a.stem && p->kh * p->kw > 1) ? tap_minor_default() : 0;
    {   // dense 1 x 1 layers (conv1 / conv3 of the bottlenecks, FC1 / FC2, their data gradients): see Conv64Args::pw_dense
        const bool in_dense = p->in_stride_w == p->cin && p->in_stride_h == (long long)p->wi * p->cin && p->in_stride_n == (long long)p->hi * p->wi * p->cin;
        const bool out_dense = p->out_stride_w == p->cout && p->out_stride_h == (long long)p->wo * p->cout && p->out_stride_n == (long long)p->ho * p->wo * p->cout;
        const long long widest = p->cin > p->cout ? p->cin : p->cout;
        const bool fits = a.M * widest < (1ll << 31), res_dense = (p->res_mode == 1 || p->res_mode == 3) && p->res_stride_w == p->out_stride_w &&
                                                                    p->res_stride_h == p->out_stride_h && p->res_stride_n == p->out_stride_n;
        a.pw_dense = 0;
        if (fits && !a.stem) {
            if (p->kh == 1 && p->kw == 1 && p->stride_h == 1 && p->stride_w == 1 && p->pad_h == 0 && p->pad_w == 0 && p->pad_mode == 0 && in_dense) a.pw_dense |= 1;
            if (out_dense) a.pw_dense |= 2;
            if (out_dense && res_dense) a.pw_dense |= 4;
        }
    }
    a.tiles_m = a.tiles_n = 0;
    a.tail_w = a.tail_b = nullptr; a.tail_deltas = a.tail_ctr = nullptr;
    a.w3 = nullptr; a.bias3 = nullptr; a.cout3 = 0; a.w3_bytes = a.out_bytes = 0;
#ifdef C64_STAMPS
    a.dbg = g_c64_stamps; a.p8 = g_p8_stamps;
#endif
    if (p->in_dtype == OSR_F16) {
        if (p->out_dtype == OSR_F16) return conv64_launch<f16_t, f16_t>(a, st);
        if (p->out_dtype == OSR_F32) return conv64_launch<f16_t, float>(a, st);
    } else {
        if (p->out_dtype == OSR_BF16) return conv64_launch<bf16_t, bf16_t>(a, st);
        if (p->out_dtype == OSR_F32) return conv64_launch<bf16_t, float>(a, st);
    }
    osr_set_error("osr_conv2d_fwd: out_dtype must equal in_dtype or be f32");
    return OSR_ERR_UNSUPPORTED;
}

// Whole ClsFreeRPNHead.forward for one level (3x3 conv + ReLU + channel L2-normalise + two 1x1 convs + sigmoid) in one
// launch; deltas/ctr are written at pixel index n*ho*wo + oh*wo + ow. Returns OSR_ERR_UNSUPPORTED when the shape is
// outside the fused kernel's envelope (the caller then runs osr_conv2d_fwd + osr_cfrpn_head_tail).
extern "C" osr_status osr_cfrpn_head_fwd_ex(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const float* w_tail,
                                            const float* b_tail, float* deltas, float* ctr, void* hidden_out, void* stream);
extern "C" osr_status osr_cfrpn_head_fwd(const osr_conv_params* p, const void* in, const void* weight, const float* bias,
                                         const float* w_tail, const float* b_tail, float* deltas, float* ctr, void* stream) {
    return osr_cfrpn_head_fwd_ex(p, in, weight, bias, w_tail, b_tail, deltas, ctr, nullptr, stream);
}

extern "C" osr_status osr_cfrpn_head_fwd_ex(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const float* w_tail,
                                            const float* b_tail, float* deltas, float* ctr, void* hidden_out, void* stream) {
    OSR_REQUIRE(p && in && weight && bias && w_tail && b_tail && deltas && ctr, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_fwd: null pointer");
    OSR_REQUIRE((((uintptr_t)hidden_out) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_fwd: hidden_out must be 16-byte aligned");
    OSR_REQUIRE(p->cout == 256 && p->cin % 64 == 0 && p->pad_mode == 0 && p->res_mode == 0, OSR_ERR_UNSUPPORTED,
                "osr_cfrpn_head_fwd: fused path needs cout == 256, cin %% 64 == 0, no residual");
    OSR_REQUIRE(p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_cfrpn_head_fwd: in_dtype must be f16/bf16");
    OSR_REQUIRE(p->n >= 1 && p->hi >= 1 && p->wi >= 1 && p->kh >= 1 && p->kw >= 1 && p->kh <= 16 && p->kw <= 16 && p->stride_h >= 1 && p->stride_w >= 1 &&
                    p->pad_h >= 0 && p->pad_w >= 0, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_fwd: bad geometry");
    OSR_REQUIRE((p->hi + 2 * p->pad_h - p->kh) / p->stride_h + 1 == p->ho && (p->wi + 2 * p->pad_w - p->kw) / p->stride_w + 1 == p->wo,
                OSR_ERR_INVALID_ARG, "osr_cfrpn_head_fwd: ho/wo inconsistent with hi/wi/kernel/stride/pad");
    OSR_REQUIRE(p->in_stride_w % 8 == 0 && p->in_stride_h % 8 == 0 && p->in_stride_n % 8 == 0 && p->in_stride_n > 0, OSR_ERR_INVALID_ARG,
                "osr_cfrpn_head_fwd: input strides must be multiples of 8 elements");
    OSR_REQUIRE((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)deltas) & 15) == 0, OSR_ERR_INVALID_ARG,
                "osr_cfrpn_head_fwd: pointers must be 16-byte aligned");
    const long long in_bytes = (long long)p->n * p->in_stride_n * 2, w_bytes = (long long)p->cout * p->kh * p->kw * p->cin * 2;
    OSR_REQUIRE(osr_conv64_eligible(p, in_bytes, w_bytes), OSR_ERR_UNSUPPORTED, "osr_cfrpn_head_fwd: tensor too large for 32-bit buffer offsets");
    Conv64Args a;
    a.p = *p; a.in = in; a.w = weight; a.bias = bias; a.res = nullptr; a.mask = nullptr; a.out = hidden_out;
    a.M = (long long)p->n * p->ho * p->wo;
    a.K = p->kh * p->kw * p->cin;
    a.div_howo = fastdiv_make((unsigned)(p->ho * p->wo));
    a.div_wo = fastdiv_make((unsigned)p->wo);
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
    a.stem = 0;
    a.tap_minor = p->kh * p->kw > 1 ? tap_minor_default() : 0;
    a.tail_w = w_tail; a.tail_b = b_tail; a.tail_deltas = deltas; a.tail_ctr = ctr;
    a.w3 = nullptr; a.bias3 = nullptr; a.cout3 = 0; a.w3_bytes = a.out_bytes = 0;
#ifdef C64_STAMPS
    a.dbg = nullptr; a.p8 = g_p8_stamps;
#endif
    hipStream_t st = (hipStream_t)stream;
    return p->in_dtype == OSR_F16 ? cfrpn_fused_launch<f16_t>(a, st) : cfrpn_fused_launch<bf16_t>(a, st);
}

// ---- two convolutions of one input in one launch (include/osr.h: osr_conv2d_fwd_pair) ---------------------------------------------
// [d2] BottleneckBlock.forward of a stage's FIRST block reads its input twice: `out = self.conv1(x)` (1x1, stride s, ReLU) and
// `shortcut = self.shortcut(x)` (1x1, stride s, no ReLU; build_resnet_fpn_backbone, Base-RCNN-FPN.yaml:3-8). Both are 1x1 layers of a few
// K slices whose time is the strided gather of x and the per-tile latency chain, not their FLOPs: as ONE grid (column groups: the N
// tiles of the shortcut, then those of conv1) the gather runs once per M tile row-block instead of in two launches, and the small conv1
// launch no longer leaves the chip half empty. Bit-identical to the two osr_conv2d_fwd launches on the 128 x 128 tile.
extern "C" osr_status osr_conv2d_fwd_pair(const osr_conv_params* p, const void* in, const void* w_a, const float* bias_a, int32_t cout_a, int32_t relu_a,
                                          void* out_a, const void* w_b, const float* bias_b, int32_t cout_b, int32_t relu_b, void* out_b, void* stream) {
    OSR_REQUIRE(p && in && w_a && bias_a && out_a && w_b && bias_b && out_b, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd_pair: null pointer");
    OSR_REQUIRE((p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16) && p->out_dtype == p->in_dtype, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd_pair: f16 / bf16 storage in and out");
    OSR_REQUIRE(p->res_mode == 0 && p->pad_mode == 0 && !p->row_seg_counts, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd_pair: no residual / stem view / row segments");
    OSR_REQUIRE(cout_a >= 128 && cout_b >= 128 && cout_a % 128 == 0 && cout_b % 128 == 0 && p->cin % 64 == 0, OSR_ERR_UNSUPPORTED,
                "osr_conv2d_fwd_pair: both output widths multiples of 128, cin a multiple of 64");
    OSR_REQUIRE(p->n >= 1 && p->hi >= 1 && p->wi >= 1 && p->kh >= 1 && p->kw >= 1 && p->kh * p->kw <= 31 && p->stride_h >= 1 && p->stride_w >= 1 && p->pad_h >= 0 && p->pad_w >= 0,
                OSR_ERR_INVALID_ARG, "osr_conv2d_fwd_pair: bad geometry");
    OSR_REQUIRE((p->hi + 2 * p->pad_h - p->kh) / p->stride_h + 1 == p->ho && (p->wi + 2 * p->pad_w - p->kw) / p->stride_w + 1 == p->wo, OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd_pair: ho/wo inconsistent with hi/wi/kernel/stride/pad");
    OSR_REQUIRE(p->in_stride_w % 8 == 0 && p->in_stride_h % 8 == 0 && p->in_stride_n % 8 == 0 && p->in_stride_n > 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd_pair: input strides must be multiples of 8 elements");
    OSR_REQUIRE((((uintptr_t)in | (uintptr_t)w_a | (uintptr_t)w_b | (uintptr_t)bias_a | (uintptr_t)bias_b | (uintptr_t)out_a | (uintptr_t)out_b) & 15) == 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd_pair: pointers must be 16-byte aligned");
    const long long M = (long long)p->n * p->ho * p->wo, K = (long long)p->kh * p->kw * p->cin, in_bytes = (long long)p->n * p->in_stride_n * 2;
    const long long wide = cout_a > cout_b ? cout_a : cout_b;
    OSR_REQUIRE(M < (1ll << 31) - 1024 && in_bytes < (1ll << 31) - 4096 && wide * K * 2 < (1ll << 31) - 4096 && M * wide < (1ll << 31), OSR_ERR_UNSUPPORTED,
                "osr_conv2d_fwd_pair: tensor too large for 32-bit offsets");
    Conv64Args a;
    a.p = *p; a.p.cout = cout_a + cout_b;
    a.in = in; a.w = w_a; a.bias = bias_a; a.res = nullptr; a.mask = nullptr; a.out = out_a;
    a.M = M; a.K = (int)K;
    a.div_howo = fastdiv_make((unsigned)(p->ho * p->wo));
    a.div_wo = fastdiv_make((unsigned)p->wo);
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)(cout_a * K * 2);
    a.stem = 0; a.tap_minor = p->kh * p->kw > 1 ? tap_minor_default() : 0;
    a.tail_w = a.tail_b = nullptr; a.tail_deltas = a.tail_ctr = nullptr; a.tail_lds_off = 0;
    a.w3 = nullptr; a.bias3 = nullptr; a.cout3 = 0; a.w3_bytes = a.out_bytes = 0;
#ifdef C64_STAMPS
    a.dbg = g_c64_stamps; a.p8 = nullptr;
#endif
    const bool in_dense = p->in_stride_w == p->cin && p->in_stride_h == (long long)p->wi * p->cin && p->in_stride_n == (long long)p->hi * p->wi * p->cin;
    a.pw_dense = 2 | ((p->kh == 1 && p->kw == 1 && p->stride_h == 1 && p->stride_w == 1 && p->pad_h == 0 && p->pad_w == 0 && in_dense) ? 1 : 0);  // both outputs are dense
    a.ngroups = 2;
    const int couts[2] = {cout_a, cout_b}, relus[2] = {relu_a, relu_b};
    const void* ws[2] = {w_a, w_b}; const float* bs[2] = {bias_a, bias_b}; void* outs[2] = {out_a, out_b};
    int tb = 0;
    for (int g = 0; g < 2; ++g) {
        Conv64Args::Group& G = a.cg[g];
        G.w = ws[g]; G.bias = bs[g]; G.out = outs[g]; G.cout = couts[g]; G.relu = relus[g] ? 1 : 0;
        G.out_stride_h = (long long)p->wo * couts[g]; G.out_stride_n = (long long)p->ho * G.out_stride_h;
        G.w_bytes = (unsigned)(couts[g] * K * 2);
        G.tile_begin = tb;
        tb += couts[g] / 128;
    }
    a.two_stage = 0;
    a.tiles_m = (int)((M + 127) / 128);
    a.tiles_n = tb;
    a.tile0 = 0; a.ntile = a.tiles_m * a.tiles_n; a.ksplit = 1; a.split_stride = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = conv64_lds_bytes(128, 128, 0, 4);
    if (p->in_dtype == OSR_F16) hipLaunchKernelGGL((conv_igemm64_kernel<f16_t, f16_t, 128, 128, 2, 2, 0, 0, 0>), dim3((unsigned)a.ntile), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_igemm64_kernel<bf16_t, bf16_t, 128, 128, 2, 2, 0, 0, 0>), dim3((unsigned)a.ntile), dim3(256), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_conv2d_fwd_pair: launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

// ---- multi-level launches (include/osr.h: osr_conv2d_fwd_levels, osr_cfrpn_head_fwd_levels) ------------------------------------
// The FPN's output convolutions ([d2] FPN.forward: `output_conv(prev_features)` per level, selected by Base-RCNN-FPN.yaml:3-8) and
// ClsFreeRPNHead.forward's `for x in features` (classification_free_rpn.py:157-161) apply ONE set of weights to every pyramid level.
// As separate launches the small levels leave most of the chip idle and every level pays its own partial last dispatch round
// (at batch 16: p2 4200 tiles of 256 x 256 = 16.4 rounds, p3 1050 = 4.1, p4 263 = 1.03, p5 66, p6 18: 25 tile times); as one
// grid of 5597 tiles it is 21.9 rounds. Every tile runs the very code of the single-level launch on its level's fields: outputs
// are bit-identical to the per-level launches on the 256 x 256 tile.
static osr_status conv64_fill_levels(Conv64Args& a, const osr_conv_params* p, int32_t nlevels, const osr_conv_level* levels, bool head, const char* who) {
    OSR_REQUIRE(nlevels >= 1 && nlevels <= C64_MAX_LEVELS && levels, OSR_ERR_INVALID_ARG, "%s: 1..%d levels", who, C64_MAX_LEVELS);
    OSR_REQUIRE(p->stride_h == 1 && p->stride_w == 1 && p->kh == p->kw && (p->kh & 1) && p->pad_h == p->kh / 2 && p->pad_w == p->kw / 2 && p->pad_mode == 0 &&
                    p->res_mode == 0 && !p->row_seg_counts, OSR_ERR_UNSUPPORTED, "%s: stride-1 same-padding convolution without residual / row segments", who);
    OSR_REQUIRE(p->cout % 256 == 0 && p->cin % 64 == 0 && p->kh * p->kw <= 31 && p->kh * p->kw * p->cin / 64 >= 2, OSR_ERR_UNSUPPORTED,
                "%s: cout %% 256 == 0, cin %% 64 == 0, at least two 64-wide K slices", who);
    long long tiles = 0;
    for (int l = 0; l < nlevels; ++l) {
        const osr_conv_level& v = levels[l];
        OSR_REQUIRE(v.in && v.n >= 1 && v.hi >= 1 && v.wi >= 1, OSR_ERR_INVALID_ARG, "%s: level %d: null input or bad size", who, l);
        OSR_REQUIRE(head ? (v.deltas && v.ctr) : v.out != nullptr, OSR_ERR_INVALID_ARG, "%s: level %d: null output", who, l);
        OSR_REQUIRE((((uintptr_t)v.in | (uintptr_t)v.out | (uintptr_t)v.deltas) & 15) == 0, OSR_ERR_INVALID_ARG, "%s: level %d: pointers must be 16-byte aligned", who, l);
        const long long rows = (long long)v.n * v.hi * v.wi, in_bytes = rows * p->cin * 2;
        OSR_REQUIRE(rows < (1ll << 31) - 1024 && in_bytes < (1ll << 31) - 4096, OSR_ERR_UNSUPPORTED, "%s: level %d too large for 32-bit buffer offsets", who, l);
        Conv64Args::Level& L = a.lv[l];
        L.in = v.in; L.out = v.out; L.tail_deltas = v.deltas; L.tail_ctr = v.ctr;
        OSR_REQUIRE((v.weight || a.w) && (v.bias || a.bias), OSR_ERR_INVALID_ARG, "%s: level %d: no weight / bias (neither its own nor a shared one)", who, l);
        OSR_REQUIRE((((uintptr_t)v.weight | (uintptr_t)v.bias) & 15) == 0, OSR_ERR_INVALID_ARG, "%s: level %d: pointers must be 16-byte aligned", who, l);
        L.w = v.weight ? v.weight : a.w; L.bias = v.bias ? v.bias : a.bias;
        L.M = rows; L.hi = v.hi; L.wi = v.wi; L.in_bytes = (unsigned)in_bytes;
        L.in_stride_h = (long long)v.wi * p->cin; L.in_stride_n = (long long)v.hi * L.in_stride_h;
        L.out_stride_h = (long long)v.wi * p->cout; L.out_stride_n = (long long)v.hi * L.out_stride_h;
        L.div_howo = fastdiv_make((unsigned)(v.hi * v.wi)); L.div_wo = fastdiv_make((unsigned)v.wi);
        L.tile_begin = (int)tiles;
        tiles += (rows + 255) / 256;
        OSR_REQUIRE(tiles * (p->cout / 256) < (1ll << 30), OSR_ERR_UNSUPPORTED, "%s: too many tiles", who);
    }
    a.nlevels = nlevels;
    a.tiles_m = (int)tiles;
    return OSR_OK;
}

extern "C" osr_status osr_conv2d_fwd_levels(const osr_conv_params* p, int32_t nlevels, const osr_conv_level* levels, const void* weight, const float* bias,
                                            void* stream) {
    OSR_REQUIRE(p, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd_levels: null pointer");
    OSR_REQUIRE((p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16) && p->out_dtype == p->in_dtype, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd_levels: f16 / bf16 storage in and out");
    OSR_REQUIRE((((uintptr_t)weight | (uintptr_t)bias) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd_levels: pointers must be 16-byte aligned");
    Conv64Args a;
    a.p = *p; a.p.in_stride_w = p->cin; a.p.out_stride_w = p->cout;
    a.in = nullptr; a.w = weight; a.bias = bias; a.res = nullptr; a.mask = nullptr; a.out = nullptr;
    a.M = 0; a.K = p->kh * p->kw * p->cin;
    a.div_howo = a.div_wo = fastdiv_make(1u);
    a.in_bytes = 0; a.w_bytes = (unsigned)((long long)p->cout * a.K * 2);
    OSR_REQUIRE((long long)p->cout * a.K * 2 < (1ll << 31) - 4096, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd_levels: weights too large for 32-bit buffer offsets");
    a.stem = 0; a.tap_minor = p->kh * p->kw > 1 ? tap_minor_default() : 0;
    a.tail_w = a.tail_b = nullptr; a.tail_deltas = a.tail_ctr = nullptr;
    a.w3 = nullptr; a.bias3 = nullptr; a.cout3 = 0; a.w3_bytes = a.out_bytes = 0;
    a.tail_lds_off = 0;
#ifdef C64_STAMPS
    a.dbg = nullptr; a.p8 = nullptr;
#endif
    const osr_status s = conv64_fill_levels(a, p, nlevels, levels, false, "osr_conv2d_fwd_levels");
    if (s != OSR_OK) return s;
    a.tiles_n = p->cout / 256;
    a.tile0 = 0; a.ntile = a.tiles_m * a.tiles_n; a.ksplit = 1; a.split_stride = 0;
    a.pw_dense = 2;  // every level's output is a dense (rows, cout) tensor
    for (int l = 0; l < nlevels; ++l)
        if (a.lv[l].M * p->cout >= (1ll << 31)) a.pw_dense = 0;
    a.two_stage = 2;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = conv64_lds_bytes(256, 256, 2, 8);
    static osr_dev_mask attr{0};
    osr_once_per_device(attr, [] {
        allow_big_lds(conv_igemm64_kernel<f16_t, f16_t, 256, 256, 2, 4, 0, 2, 0>);
        allow_big_lds(conv_igemm64_kernel<bf16_t, bf16_t, 256, 256, 2, 4, 0, 2, 0>);
    });
    if (p->in_dtype == OSR_F16) hipLaunchKernelGGL((conv_igemm64_kernel<f16_t, f16_t, 256, 256, 2, 4, 0, 2, 0>), dim3((unsigned)a.ntile), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((conv_igemm64_kernel<bf16_t, bf16_t, 256, 256, 2, 4, 0, 2, 0>), dim3((unsigned)a.ntile), dim3(512), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_conv2d_fwd_levels: launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

template <class TI>
static osr_status cfrpn_levels_launch(Conv64Args& a, hipStream_t st) {
    a.two_stage = 1;
    a.tiles_n = 1; a.tile0 = 0; a.ntile = a.tiles_m; a.ksplit = 1; a.split_stride = 0;
    const size_t t_bytes = (size_t)256 * (256 + 8) * 2, stages = (size_t)2 * (256 + 256) * 128;
    a.tail_lds_off = (int)(t_bytes > stages ? t_bytes : stages);
    const size_t lds = (size_t)a.tail_lds_off + C64_TAIL_LDS(256) + 32;
    static osr_dev_mask attr{0};
    osr_once_per_device(attr, [] { allow_big_lds(conv_igemm64_kernel<TI, TI, 256, 256, 2, 4, 1, 2>); });
    hipLaunchKernelGGL((conv_igemm64_kernel<TI, TI, 256, 256, 2, 4, 1, 2>), dim3((unsigned)a.tiles_m), dim3(512), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_cfrpn_head_fwd_levels: launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

extern "C" osr_status osr_cfrpn_head_fwd_levels(const osr_conv_params* p, int32_t nlevels, const osr_conv_level* levels, const void* weight, const float* bias,
                                                const float* w_tail, const float* b_tail, void* stream) {
    OSR_REQUIRE(p && weight && bias && w_tail && b_tail, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_fwd_levels: null pointer");
    OSR_REQUIRE(p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_cfrpn_head_fwd_levels: in_dtype must be f16/bf16");
    OSR_REQUIRE(p->cout == 256, OSR_ERR_UNSUPPORTED, "osr_cfrpn_head_fwd_levels: cout must be 256");
    OSR_REQUIRE((((uintptr_t)weight | (uintptr_t)bias) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_fwd_levels: pointers must be 16-byte aligned");
    Conv64Args a;
    a.p = *p; a.p.in_stride_w = p->cin; a.p.out_stride_w = p->cout;
    a.in = nullptr; a.w = weight; a.bias = bias; a.res = nullptr; a.mask = nullptr; a.out = nullptr;
    a.M = 0; a.K = p->kh * p->kw * p->cin;
    a.div_howo = a.div_wo = fastdiv_make(1u);
    a.in_bytes = 0; a.w_bytes = (unsigned)((long long)p->cout * a.K * 2);
    a.stem = 0; a.tap_minor = p->kh * p->kw > 1 ? tap_minor_default() : 0;
    a.tail_w = w_tail; a.tail_b = b_tail; a.tail_deltas = a.tail_ctr = nullptr;
    a.w3 = nullptr; a.bias3 = nullptr; a.cout3 = 0; a.w3_bytes = a.out_bytes = 0;
#ifdef C64_STAMPS
    a.dbg = nullptr; a.p8 = nullptr;
#endif
    const osr_status s = conv64_fill_levels(a, p, nlevels, levels, true, "osr_cfrpn_head_fwd_levels");
    if (s != OSR_OK) return s;
    OSR_REQUIRE(a.K / 64 >= 8, OSR_ERR_UNSUPPORTED, "osr_cfrpn_head_fwd_levels: at least eight K slices");
    hipStream_t st = (hipStream_t)stream;
    return p->in_dtype == OSR_F16 ? cfrpn_levels_launch<f16_t>(a, st) : cfrpn_levels_launch<bf16_t>(a, st);
}

// A bottleneck's conv2 -> conv3 in one launch (include/osr.h: osr_conv2d_chain_fwd):
//     out = relu(conv1x1(act(conv(in, w) + bias), w3) + bias3 + residual),   act = ReLU when p->relu.
// p describes the FIRST convolution (cout == 128; its output never reaches HBM); out and residual are dense (rows, cout3) tensors.
template <class TI, int BM>
static osr_status conv_chain_launch(Conv64Args& a, hipStream_t st) {
    constexpr int WM = BM / 64;
    a.two_stage = 1;
    a.tiles_n = 1;
    a.tiles_m = (int)((a.M + BM - 1) / BM);
    a.tile0 = 0; a.ntile = a.tiles_m; a.ksplit = 1; a.split_stride = 0;
    // parked tile + three weight stages; the K loop's two staging buffers alias the front
    const size_t park = (size_t)2 * BM * 128 + 3 * 16384, stg = (size_t)2 * (BM + 128) * 128, lds = park > stg ? park : stg;
    static osr_dev_mask attr{0};
    osr_once_per_device(attr, [] { allow_big_lds(conv_igemm64_kernel<TI, TI, BM, 128, WM, 2, 2, 1>); });
    hipLaunchKernelGGL((conv_igemm64_kernel<TI, TI, BM, 128, WM, 2, 2, 1>), dim3((unsigned)a.tiles_m), dim3(WM * 2 * 64), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_conv2d_chain_fwd: launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

extern "C" osr_status osr_conv2d_chain_fwd_ex(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* w3,
                                              const float* bias3, int32_t cout3, const void* residual, void* out, void* mid_out, void* stream);
extern "C" osr_status osr_conv2d_chain_fwd(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* w3,
                                           const float* bias3, int32_t cout3, const void* residual, void* out, void* stream) {
    return osr_conv2d_chain_fwd_ex(p, in, weight, bias, w3, bias3, cout3, residual, out, nullptr, stream);
}

extern "C" osr_status osr_conv2d_chain_fwd_ex(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* w3,
                                              const float* bias3, int32_t cout3, const void* residual, void* out, void* mid_out, void* stream) {
    OSR_REQUIRE(p && in && weight && bias && w3 && bias3 && residual && out, OSR_ERR_INVALID_ARG, "osr_conv2d_chain_fwd: null pointer");
    OSR_REQUIRE((((uintptr_t)mid_out) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_conv2d_chain_fwd: mid_out must be 16-byte aligned");
    OSR_REQUIRE(p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_conv2d_chain_fwd: in_dtype must be f16/bf16");
    OSR_REQUIRE(p->out_dtype == p->in_dtype, OSR_ERR_UNSUPPORTED, "osr_conv2d_chain_fwd: out_dtype must equal in_dtype");
    OSR_REQUIRE(p->cout == 128 && cout3 == 512 && p->cin % 64 == 0 && p->pad_mode == 0, OSR_ERR_UNSUPPORTED,
                "osr_conv2d_chain_fwd: needs cout == 128, cout3 == 512, cin %% 64 == 0 (got %d, %d, %d)", p->cout, cout3, p->cin);
    OSR_REQUIRE(p->n >= 1 && p->hi >= 1 && p->wi >= 1 && p->kh >= 1 && p->kw >= 1 && p->kh <= 16 && p->kw <= 16 && p->stride_h >= 1 && p->stride_w >= 1 &&
                    p->pad_h >= 0 && p->pad_w >= 0, OSR_ERR_INVALID_ARG, "osr_conv2d_chain_fwd: bad geometry");
    OSR_REQUIRE((p->hi + 2 * p->pad_h - p->kh) / p->stride_h + 1 == p->ho && (p->wi + 2 * p->pad_w - p->kw) / p->stride_w + 1 == p->wo,
                OSR_ERR_INVALID_ARG, "osr_conv2d_chain_fwd: ho/wo inconsistent with hi/wi/kernel/stride/pad");
    OSR_REQUIRE(p->in_stride_w % 8 == 0 && p->in_stride_h % 8 == 0 && p->in_stride_n % 8 == 0 && p->in_stride_n > 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_chain_fwd: input strides must be multiples of 8 elements");
    OSR_REQUIRE((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)w3 | (uintptr_t)bias3 | (uintptr_t)residual | (uintptr_t)out) & 15) == 0,
                OSR_ERR_INVALID_ARG, "osr_conv2d_chain_fwd: pointers must be 16-byte aligned");
    const long long in_bytes = (long long)p->n * p->in_stride_n * 2, w_bytes = (long long)p->cout * p->kh * p->kw * p->cin * 2;
    const long long M = (long long)p->n * p->ho * p->wo, out_bytes = M * cout3 * 2;
    OSR_REQUIRE(osr_conv64_eligible(p, in_bytes, w_bytes) && out_bytes < (1ll << 31) - 4096, OSR_ERR_UNSUPPORTED,
                "osr_conv2d_chain_fwd: tensor too large for 32-bit buffer offsets");
    Conv64Args a;
    a.p = *p; a.in = in; a.w = weight; a.bias = bias; a.res = residual; a.mask = nullptr; a.out = out;
    a.p.row_seg_counts = nullptr;
    a.M = M;
    a.K = p->kh * p->kw * p->cin;
    a.div_howo = fastdiv_make((unsigned)(p->ho * p->wo));
    a.div_wo = fastdiv_make((unsigned)p->wo);
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
    a.stem = 0;
    a.tap_minor = p->kh * p->kw > 1 ? tap_minor_default() : 0;
    a.tail_w = a.tail_b = nullptr; a.tail_deltas = a.tail_ctr = nullptr; a.tail_lds_off = 0;
    a.w3 = w3; a.bias3 = bias3; a.cout3 = cout3;
    a.w3_bytes = (unsigned)((long long)cout3 * p->cout * 2); a.out_bytes = (unsigned)out_bytes;
    a.mid_out = mid_out;
#ifdef C64_STAMPS
    a.dbg = g_c64_stamps; a.p8 = g_p8_stamps;
#endif
    hipStream_t st = (hipStream_t)stream;
    return p->in_dtype == OSR_F16 ? conv_chain_launch<f16_t, 128>(a, st) : conv_chain_launch<bf16_t, 128>(a, st);
}
