// Backward kernels of the dense layers (training step, backward half).
//
//  * backward-data (dgrad) needs no kernel of its own: it is a forward convolution of dy with the spatially flipped,
//    transposed weights (host/weights.py pack_dgrad_weight) through osr_conv2d_fwd -- stride-2 1x1 layers write every
//    second pixel of a zeroed dx through the output strides, and the ReLU mask / gradient sum of the layer below ride in
//    the epilogue (res_mode 3 / 1).
//  * backward-weights (wgrad):  dw[co][kh][kw][ci] = sum_m dy[m][co] * x[n, oh*sh-ph+kh, ow*sw-pw+kw, ci],  m = (n,oh,ow).
//    The contraction index m is the ROW index of both operands in memory, so both MFMA operands are "k-strided". Tiles
//    are staged row-major ([m][128 channels], 256-B rows, LDS-DMA with a source-side XOR swizzle of the 16-B chunks) and
//    read with gfx950's transposing ds_read_b64_tr_b16: one read hands lane (g = lane>>4, c = lane&15) channel c of rows
//    8g+4h .. 8g+4h+3, i.e. exactly half of its 16x16x32 fragment (k = 8g .. 8g+7), in natural k order.
//    One workgroup = one 128(co) x 128(ci) tile of one tap over one M chunk (split-K); fp32 partials go to a workspace and
//    a second kernel sums them in a fixed order (bitwise reproducible, no atomics).
#include "osr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <class T> struct FragW;
template <> struct FragW<f16_t> {
    typedef f16x8 type;
    static __device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct FragW<bf16_t> {
    typedef bf16x8 type;
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

struct WDiv {
    unsigned mp, sh1, sh2, d;
};
static WDiv wdiv_make(unsigned d) {
    WDiv f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mp = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    f.d = d;
    return f;
}
__device__ __forceinline__ unsigned wdiv(unsigned n, const WDiv& f) {
    const unsigned t = __umulhi(f.mp, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct WgradArgs {
    osr_conv_params p;
    const void* x;
    const void* dy;
    float* partial;       // [splits][cout][kh*kw][cin]
    long long M;          // n*ho*wo
    long long rows_per_split;  // multiple of 64
    int splits, tiles_co, tiles_ci, taps;
    unsigned x_bytes, dy_bytes;
    WDiv div_howo, div_wo;
};

#define WG_OOB 0x80000000u
#ifndef WG_BM
#define WG_BM 64  // rows (m) per pipeline step
#endif
#ifndef WG_P8_MIN_STEPS
#define WG_P8_MIN_STEPS 48  // 64-row steps per split from which the 256 x 256 tile runs the 8-phase loop (measured: scripts/exp_wgrad8.py)
#endif
#ifndef WG_NST
#define WG_NST 2  // staging buffers: the pieces of step s + WG_NST - 1 are in flight under step s
#endif

// Tile TCO (output channels) x TCI (input channels) of one tap, WM x WN waves. Per step of 64 pixel rows the dy tile
// [64][TCO] and the x tile [64][TCI] are staged row-major (LDS-DMA pieces of 1 KiB = 512/T rows); the 16-byte chunk c of row r
// lands in slot c ^ sw(r), sw(r) = 2 ((r & 3) | ((r >> 3) & 1) << 2): the eight rows one half-wave touches in a transposing read
// (r = 8g + 4h + q, g in {0,1} or {2,3}) get eight different EVEN swizzles. Even, because the lanes of a read differ in the low chunk
// bit as well (pp >> 1: the two 16-byte chunks of a 16-channel sub-tile): with the odd values of round 1's sw(r) = (r & 3) | ... lane
// (row s, chunk bit 0) and lane (row s ^ 1, chunk bit 1) met in the same banks -- rocprofv3: SQ_LDS_BANK_CONFLICT = half of
// SQ_LDS_IDX_ACTIVE in both instantiations. Now the 32 lanes of a half-wave cover all 64 banks once.
// The pieces of step s+1 are issued between the MFMAs of step s (as in the forward kernel).
template <class TI, int TCO, int TCI, int WM, int WN, int P8 = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int NW = WM * WN;
    constexpr int SA = TCO / WM / 16, SB = TCI / WN / 16;       // 16-wide sub-tiles per wave
    constexpr int YB = TCO * 2, XB = TCI * 2;                   // bytes per staged row
    constexpr int STAGE = WG_BM * (YB + XB);
    constexpr int YP = WG_BM * YB / 1024, XP = WG_BM * XB / 1024;  // pieces per step
    constexpr int PPW = (YP + XP) / NW;                         // pieces per wave and step
    static_assert((YP + XP) % NW == 0 && YP % NW == 0, "pieces must divide over the waves, dy pieces first");
    constexpr int YPW = YP / NW, XPW = XP / NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // WG_NST stages
    typedef typename FragW<TI>::type frag_t;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const osr_conv_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;

    // tile decode: consecutive work items share the M chunk (same dy / x rows), then the dy columns. Round 6: workgroups b and b + 8 share an
    // XCD (its L2), so CONSECUTIVE workgroup ids land on eight different L2s -- the nine taps of a 3 x 3 layer (or the ci / co tiles of a
    // 1 x 1 layer) that read the same rows each fetched them into an L2 of their own, and the rows came from the Infinity Cache / HBM up
    // to nine times. The same bijective remap as the forward kernel's: every XCD walks one contiguous run of the linear work index, so the
    // tiles of a row chunk run on ONE XCD, back to back (WG_XCD_REMAP 0: the old order, for A/B; partial sums land in the same slots either way)
#ifndef WG_XCD_REMAP
#define WG_XCD_REMAP 1
#endif
    const int ntiles = a.tiles_co * a.tiles_ci * a.taps;
    int bid = (int)blockIdx.x;
    if (WG_XCD_REMAP) {
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / ntiles;
    int t = bid - split * ntiles;
    const int tile_ci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tap = t % a.taps;
    const int tile_co = t / a.taps;
    const int kh = tap / p.kw, kw = tap - kh * p.kw;
    const int co0 = tile_co * TCO, ci0 = tile_ci * TCI;
    const long long m_begin = (long long)split * a.rows_per_split;
    const long long m_end = m_begin + a.rows_per_split < a.M ? m_begin + a.rows_per_split : a.M;
    const int nsteps = (int)((m_end - m_begin + WG_BM - 1) / WG_BM);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);

    // ---- staging addresses, kept INCREMENTALLY. A lane serves the same row-in-step and chunk of every step; from one step to the next
    // its pixel row advances by WG_BM. Round 1-2 recomputed (n, oh, ow) of every piece with two magic-number divisions and three
    // 64-bit multiplies per piece and step, inside the MFMA loop: 163 quarter-rate v_mul_lo_u32 and ~150 exec-masked branches in the
    // loop body, 3.1 other vector instructions per MFMA (rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA). Now: the lane's FIRST piece keeps
    // (ih, iw, byte offset of its tap pixel) and advances them by WG_BM rows per step with adds and wraps; the wave's other pieces
    // sit a fixed number of rows further on and are derived with adds and one wrap. No division, no 32-bit multiply in the loop. ----
#define WG_SW(r) ((((r) & 3) | ((((r) >> 3) & 1) << 2)) << 1)
    constexpr int CPRY = YB / 16, RPPY = 1024 / YB, CPRX = XB / 16, RPPX = 1024 / XB;
    static_assert(!P8 || (TCO == 256 && TCI == 256 && WM == 2 && WN == 4 && WG_BM == 64 && WG_NST == 2), "8-phase loop: 256 x 256 tile, 2 x 4 waves, 64-row steps, two stages");
    // P8 (the 8-phase loop below): a step is staged in four UNITS of 32 pixel rows (dy rows / x rows of K half kk), 16 pieces each, two per
    // wave: the wave's piece jj of K half kk holds rows kk * 32 + wid * 4 + jj * 2 + {0, 1}. Otherwise a wave's pieces are consecutive.
    const int rowy0 = P8 ? wid * 4 + lane / CPRY : wid * YPW * RPPY + lane / CPRY;
    const int rowx0 = P8 ? wid * 4 + lane / CPRX : wid * XPW * RPPX + lane / CPRX;  // piece q: + (row offset of piece q) rows
    auto piece_rows = [&](int q, int rpp) -> int { return P8 ? (q >> 1) * 32 + (q & 1) * 2 : q * rpp; };  // rows between the lane's first piece and piece q
    auto piece_slot = [&](int q, int ppw) -> int { return P8 ? (q >> 1) * 16 + wid * 2 + (q & 1) : wid * ppw + q; };  // 1-KiB slot of piece q in the tile
    // dy: offset of (row m, co0) grows by cout elements per row
    unsigned ybase = (unsigned)(((m_begin + rowy0) * p.cout + co0) * 2);
    const unsigned ystep = (unsigned)((long long)WG_BM * p.cout * 2), yrow = (unsigned)(p.cout * 2);
    long long my = m_begin + rowy0, mx = m_begin + rowx0;  // pixel rows of the first pieces at the current step
    // x: input coordinates (ih, iw) of pixel mx's tap (kh, kw) and their byte offset at channel ci0 (wraps below zero in the padding).
    // Output column + 1 = iw + stride_w; past the last output column (iw >= iw_lim) the next output row starts: iw -= wo * stride_w,
    // ih += stride_h; past the last output row the next image.
    int xih, xiw;
    unsigned xbase;
    {
        const long long mm = mx < a.M ? mx : 0;
        const unsigned mu = (unsigned)mm, ni = wdiv(mu, a.div_howo), rem = mu - ni * a.div_howo.d;
        const int oh0 = (int)wdiv(rem, a.div_wo), ow0 = (int)(rem - (unsigned)oh0 * a.div_wo.d);
        xih = oh0 * p.stride_h - p.pad_h + kh; xiw = ow0 * p.stride_w - p.pad_w + kw;
        xbase = (unsigned)(((long long)ni * p.in_stride_n + (long long)xih * p.in_stride_h + (long long)xiw * p.in_stride_w + ci0) * 2);
    }
    const unsigned x_dw = (unsigned)(p.stride_w * p.in_stride_w * 2);                                     // ow + 1
    const unsigned x_wrap_w = (unsigned)((p.stride_h * p.in_stride_h - (long long)p.wo * p.stride_w * p.in_stride_w) * 2);  // ow -= wo, oh + 1
    const unsigned x_wrap_h = (unsigned)((p.in_stride_n - (long long)p.ho * p.stride_h * p.in_stride_h) * 2);              // oh -= ho, image + 1
    const int iw_lim = p.wo * p.stride_w - p.pad_w + kw, ih_lim = p.ho * p.stride_h - p.pad_h + kh;
    const int iw_span = p.wo * p.stride_w, ih_span = p.ho * p.stride_h;
    // P8: every one of the lane's four x pieces (rows +0, +2, +32, +34) keeps its own (ih, iw, offset) and moves on by WG_BM rows per step in
    // CLOSED FORM -- WG_BM = dn images + dh output rows + dw output columns, then at most one column wrap and one row wrap. (The while-loop
    // wraps of the old loop take wo-dependent trip counts: with one pixel per row -- the FC layers' (m, 1) view -- 64 trips per step and up to
    // 34 per piece, which sat in this loop's load sections: FC1's weight gradient ran 2.5x slower.)
    int xih4[P8 ? XPW : 1], xiw4[P8 ? XPW : 1];
    unsigned xoff4[P8 ? XPW : 1];
    int adv_iw = 0, adv_ih = 0;
    unsigned adv_off = 0;
    {
        const unsigned dn = wdiv((unsigned)WG_BM, a.div_howo), rem = (unsigned)WG_BM - dn * a.div_howo.d;
        const unsigned dh = wdiv(rem, a.div_wo), dw = rem - dh * a.div_wo.d;
        adv_iw = (int)dw * p.stride_w; adv_ih = (int)dh * p.stride_h;
        adv_off = (unsigned)(((long long)dn * p.in_stride_n + (long long)adv_ih * p.in_stride_h + (long long)adv_iw * p.in_stride_w) * 2);
    }
    if constexpr (P8) {
#pragma unroll
        for (int q = 0; q < XPW; ++q) {
            const long long mq = mx + piece_rows(q, RPPX);
            const long long mm = mq < a.M ? mq : 0;
            const unsigned mu = (unsigned)mm, ni = wdiv(mu, a.div_howo), rm = mu - ni * a.div_howo.d;
            const int oh0 = (int)wdiv(rm, a.div_wo), ow0 = (int)(rm - (unsigned)oh0 * a.div_wo.d);
            xih4[q] = oh0 * p.stride_h - p.pad_h + kh; xiw4[q] = ow0 * p.stride_w - p.pad_w + kw;
            xoff4[q] = (unsigned)(((long long)ni * p.in_stride_n + (long long)xih4[q] * p.in_stride_h + (long long)xiw4[q] * p.in_stride_w + ci0) * 2);
        }
    }

    // one staging piece: q < YPW -> dy rows, else x rows (gathered through the conv geometry)
#define WG_PIECE(stage, step, q)                                                                                                \
    {                                                                                                                           \
        unsigned char* sy_ = lds + (stage) * STAGE;                                                                             \
        if ((q) < YPW) {                                                                                                        \
            const int qq_ = (q) < YPW ? (q) : 0;                                                                                \
            const int pc_ = piece_slot(qq_, YPW);                                                                               \
            const int dr_ = piece_rows(qq_, RPPY);                                                                              \
            const int row_ = rowy0 + dr_;                                                                                       \
            const int chunk_ = (lane % CPRY) ^ WG_SW(row_);                                                                     \
            const unsigned yoff_ = (my + dr_ < m_end && co0 + chunk_ * 8 < p.cout) ? ybase + (unsigned)dr_ * yrow + (unsigned)chunk_ * 16u : WG_OOB; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_t*)(sy_ + pc_ * 1024), 16, yoff_, 0, 0, 0);                  \
        } else {                                                                                                                \
            const int qq_ = (q) >= YPW ? (q) - YPW : 0;                                                                         \
            const int pc_ = piece_slot(qq_, XPW);                                                                               \
            const int dr_ = piece_rows(qq_, RPPX);                                                                              \
            const int row_ = rowx0 + dr_;                                                                                       \
            const int chunk_ = (lane % CPRX) ^ WG_SW(row_);                                                                     \
            int iw_ = P8 ? xiw4[P8 ? qq_ : 0] : xiw + dr_ * p.stride_w, ih_ = P8 ? xih4[P8 ? qq_ : 0] : xih;                    \
            unsigned off_ = P8 ? xoff4[P8 ? qq_ : 0] : xbase + (unsigned)dr_ * x_dw;                                            \
            if constexpr (!P8) {                                                                                                \
                while (iw_ >= iw_lim) { iw_ -= iw_span; ih_ += p.stride_h; off_ += x_wrap_w; if (ih_ >= ih_lim) { ih_ -= ih_span; off_ += x_wrap_h; } } \
            }                                                                                                                   \
            const bool ok_ = mx + dr_ < m_end && ci0 + chunk_ * 8 < p.cin && (unsigned)ih_ < (unsigned)p.hi && (unsigned)iw_ < (unsigned)p.wi; \
            const unsigned xoff_ = ok_ ? off_ + (unsigned)chunk_ * 16u : WG_OOB;                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(sy_ + WG_BM * YB + pc_ * 1024), 16, xoff_, 0, 0, 0);    \
        }                                                                                                                       \
    }
    // the state moves on by one step (WG_BM rows): called once all pieces of a step have been issued
#define WG_ADVANCE()                                                                                                            \
    {                                                                                                                           \
        my += WG_BM; mx += WG_BM; ybase += ystep;                                                                               \
        if constexpr (P8) {                                                                                                     \
            _Pragma("unroll") for (int q_ = 0; q_ < XPW; ++q_) {                                                                \
                int iw_ = xiw4[P8 ? q_ : 0] + adv_iw, ih_ = xih4[P8 ? q_ : 0] + adv_ih;                                         \
                unsigned off_ = xoff4[P8 ? q_ : 0] + adv_off;                                                                   \
                const bool cw_ = iw_ >= iw_lim;                                                                                 \
                iw_ -= cw_ ? iw_span : 0; ih_ += cw_ ? p.stride_h : 0; off_ += cw_ ? x_wrap_w : 0u;                            \
                const bool ch_ = ih_ >= ih_lim;                                                                                 \
                ih_ -= ch_ ? ih_span : 0; off_ += ch_ ? x_wrap_h : 0u;                                                          \
                xiw4[P8 ? q_ : 0] = iw_; xih4[P8 ? q_ : 0] = ih_; xoff4[P8 ? q_ : 0] = off_;                                    \
            }                                                                                                                   \
        } else {  /* the same closed form on the single state of the one-barrier loop (its pieces sit <= 6 rows further on: short while loops) */ \
            xiw += adv_iw; xih += adv_ih; xbase += adv_off;                                                                     \
            const bool cw_ = xiw >= iw_lim;                                                                                     \
            xiw -= cw_ ? iw_span : 0; xih += cw_ ? p.stride_h : 0; xbase += cw_ ? x_wrap_w : 0u;                                \
            const bool ch_ = xih >= ih_lim;                                                                                     \
            xih -= ch_ ? ih_span : 0; xbase += ch_ ? x_wrap_h : 0u;                                                             \
        }                                                                                                                       \
    }

    f32x4 acc[SA][SB];  // [co sub-tile][ci sub-tile]: rows = output channels, columns = input channels
#pragma unroll
    for (int i = 0; i < SA; ++i)
#pragma unroll
        for (int j = 0; j < SB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing fragment read: lane (g, q, pp) of a 16-lane group supplies the address of row 8g+4h+q, 4 elements at
    // column 16*sub + 4*pp of the wave's channels; it receives column (lane & 15) of the four rows.
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    constexpr int NMF = (WG_BM / 32) * SA * SB, PSTEP = (NMF / 2) / PPW > 0 ? (NMF / 2) / PPW : 1;  // pieces go out during the first half of a step
    // (pieces are issued for every step up to nsteps + WG_NST - 2: rows past m_end are zero fills into buffers nothing reads, and the
    // counted wait below stays exact)
    if constexpr (P8) {
        // ---- 8-phase loop (round 5; the forward kernel's schedule, osr_conv_gemm64.hip, with the units cut along the contraction axis: here
        // BOTH operands are k-strided). A 64-row step is four phases of 16 MFMAs, (kk, qa) = (0,0) (0,1) (1,0) (1,1): K half kk (32 pixel
        // rows) x output-channel half qa (64 of the wave's 128) x all of its 64 input channels. Units of a step: X0 Y0 X1 Y1 (x / dy rows
        // of K half 0 / 1), read in phase 1 (X0, and the Y0 columns of qa 0), 2 (Y0 columns of qa 1), 3 (X1, Y1), 4 (Y1) -- each is
        // restaged in the phase after its last read: X0 of step s+2 in phase 2, Y0 in 3, X1 in 4, Y1 in phase 1 of the next step. Every
        // phase retires its own fragment reads (lgkmcnt(0)) in front of its first barrier; waves 4-7 run one barrier behind waves 0-3.
        // Waits on the vector-memory counter: phases 2 and 4, vmcnt(10) -- FIVE units stay in flight; phase 4's wait retires the two K-half-0
        // units of the next step (read from phase 1 on), phase 2's the two K-half-1 units of this step (read from phase 3 on): a read
        // sits one phase behind the wait + barrier that retire its data. Steps past the last are zero-fill dummies (rows >= m_end). The
        // accumulation order per output element is the old loop's: bit-identical results.
        // The transposing reads are issued as inline asm: behind the builtin, hipcc (ROCm 7.2) waits vmcnt(0) in front of every group of
        // reads while an LDS-DMA is in flight (it cannot prove the read does not alias the DMA's destination), which would drain the ring
        // four times per step. The asm form leaves the ordering to this loop's own waits: lgkmcnt(0) in front of each phase's first barrier
        // (one asm statement with the barrier) and a sched_barrier in front of the MFMAs that consume the registers.
        // LDS byte address of (row g*8 + q4 [+ kk*32 + h*4], chunk c ^ sw, half pp & 1): sw depends on the lane only (row & 3 = q4, (row >> 3) & 1 = g & 1).
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)lds;
        const int sw8 = WG_SW(g * 8 + q4);
        const unsigned lane_row = (unsigned)((g * 8 + q4) * YB + (pp & 1) * 8);
        static_assert(YB == XB, "one row pitch");
        unsigned a_addr[2][SA / 2], b_addr[SB];  // [qa][i], [j]: stage 0, kk 0, h 0
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int i = 0; i < SA / 2; ++i) a_addr[qa][i] = lds0 + lane_row + (unsigned)((((wr * SA + qa * (SA / 2) + i) * 2 + (pp >> 1)) ^ sw8) * 16);
#pragma unroll
        for (int j = 0; j < SB; ++j) b_addr[j] = lds0 + (unsigned)(WG_BM * YB) + lane_row + (unsigned)((((wc * SB + j) * 2 + (pp >> 1)) ^ sw8) * 16);
        u32x2 va[2][SA / 2], vb[2][SB];
        frag_t fa[SA / 2], fb[SB];
#define W8_TR(dst_, addr_, off_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
#define W8_READ_A(kk_, qa_)                                                                                                         \
        {                                                                                                                           \
            _Pragma("unroll") for (int i = 0; i < SA / 2; ++i) {                                                                    \
                const unsigned ad_ = a_addr[qa_][i] + stage_off;                                                                    \
                W8_TR(va[0][i], ad_, (kk_) * 32 * YB);                                                                              \
                W8_TR(va[1][i], ad_, (kk_) * 32 * YB + 4 * YB);                                                                     \
            }                                                                                                                       \
        }
#define W8_READ_B(kk_)                                                                                                              \
        {                                                                                                                           \
            _Pragma("unroll") for (int j = 0; j < SB; ++j) {                                                                        \
                const unsigned ad_ = b_addr[j] + stage_off;                                                                         \
                W8_TR(vb[0][j], ad_, (kk_) * 32 * XB);                                                                              \
                W8_TR(vb[1][j], ad_, (kk_) * 32 * XB + 4 * XB);                                                                     \
            }                                                                                                                       \
        }
#define W8_ISSUE(stage_, q0_)                                                                                                       \
        {                                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            WG_PIECE(stage_, 0, (q0_));                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            WG_PIECE(stage_, 0, (q0_) + 1);                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }
#define W8_MFMA(qa_, NEWB, MID)                                                                                                     \
        {                                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));                                                            \
            _Pragma("unroll") for (int i = 0; i < SA / 2; ++i) {                                                                    \
                const u32x4_ ta = {va[0][i][0], va[0][i][1], va[1][i][0], va[1][i][1]};                                             \
                fa[i] = __builtin_bit_cast(frag_t, ta);                                                                             \
            }                                                                                                                       \
            if (NEWB) {                                                                                                             \
                _Pragma("unroll") for (int j = 0; j < SB; ++j) {                                                                    \
                    const u32x4_ tb = {vb[0][j][0], vb[0][j][1], vb[1][j][0], vb[1][j][1]};                                         \
                    fb[j] = __builtin_bit_cast(frag_t, tb);                                                                         \
                }                                                                                                                   \
            }                                                                                                                       \
            __builtin_amdgcn_s_setprio(1);                                                                                          \
            _Pragma("unroll") for (int i = 0; i < SA / 2; ++i) {                                                                    \
                _Pragma("unroll") for (int j = 0; j < SB; ++j)                                                                      \
                    acc[(qa_) * (SA / 2) + i][j] = FragW<TI>::mfma(fa[i], fb[j], acc[(qa_) * (SA / 2) + i][j]);                     \
                if (i == SA / 2 - 2) { MID; }                                                                                       \
            }                                                                                                                       \
            __builtin_amdgcn_s_setprio(0);                                                                                          \
            __builtin_amdgcn_s_barrier();                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }
#define W8_WAIT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define W8_WAITVM_BARRIER() asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory")
        // unit -> first piece index of WG_PIECE: dy pieces 0 .. YPW-1 (K half 0: 0, 1; K half 1: 2, 3), x pieces YPW .. YPW+XPW-1
        constexpr int UY0 = 0, UY1 = 2, UX0 = YPW, UX1 = YPW + 2;
        static_assert(YPW == 4 && XPW == 4, "two pieces per wave and unit");
        if (nsteps > 0) {
            // prologue: step 0 complete + X0, Y0, X1 of step 1; phase 1 of step 0 issues Y1 of step 1 (the steady-state slot of that unit)
            W8_ISSUE(0, UX0); W8_ISSUE(0, UY0); W8_ISSUE(0, UX1); W8_ISSUE(0, UY1);
            WG_ADVANCE();
            W8_ISSUE(1, UX0); W8_ISSUE(1, UY0); W8_ISSUE(1, UX1);
            asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");  // step 0 has landed, for every wave
            if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one barrier behind
            __builtin_amdgcn_sched_barrier(0);
            for (int s = 0; s < nsteps; ++s) {
                const int cs = s & 1, ns = cs ^ 1;
                const unsigned stage_off = (unsigned)(cs * STAGE);
                // phase 1: K half 0, output channels half 0 (the staging state is one step ahead: step s+1; it moves to s+2 inside the cluster)
                W8_READ_B(0);
                W8_READ_A(0, 0);
                W8_ISSUE(ns, UY1);
                W8_WAIT_BARRIER();
                W8_MFMA(0, true, { __builtin_amdgcn_sched_barrier(0); WG_ADVANCE(); __builtin_amdgcn_sched_barrier(0); });
                // phase 2: K half 0, half 1
                W8_READ_A(0, 1);
                W8_ISSUE(cs, UX0);
                W8_WAITVM_BARRIER();  // X1, Y1 of this step have landed (read from phase 3 on)
                W8_MFMA(1, false, {});
                // phase 3: K half 1, half 0
                W8_READ_B(1);
                W8_READ_A(1, 0);
                W8_ISSUE(cs, UY0);
                W8_WAIT_BARRIER();
                W8_MFMA(0, true, {});
                // phase 4: K half 1, half 1
                W8_READ_A(1, 1);
                W8_ISSUE(cs, UX1);
                W8_WAITVM_BARRIER();  // X0, Y0 of step s+1 have landed (read from its phase 1 on)
                W8_MFMA(1, false, {});
            }
            if (wr == 0) __builtin_amdgcn_s_barrier();
        }
#undef W8_READ_A
#undef W8_TR
#undef W8_READ_B
#undef W8_ISSUE
#undef W8_MFMA
#undef W8_WAIT_BARRIER
#undef W8_WAITVM_BARRIER
    } else {
    if (nsteps > 0) {
#pragma unroll
        for (int ps = 0; ps < WG_NST - 1; ++ps) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) WG_PIECE(ps, ps, q);
            WG_ADVANCE();
        }
    }
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WG_NST - 2) * PPW) : "memory");
        __syncthreads();
        const bool more = true;
        const unsigned char* sy = lds + (s % WG_NST) * STAGE;
        const unsigned char* sx = sy + WG_BM * YB;
        const int nst = (s + WG_NST - 1) % WG_NST;
#pragma unroll
        for (int kk = 0; kk < WG_BM / 32; ++kk) {
            s16x4 va[2][SA], vb[2][SB];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = kk * 32 + g * 8 + h * 4 + q4;
                const int sw = WG_SW(row);
#pragma unroll
                for (int i = 0; i < SA; ++i) {
                    const int chunk = ((wr * SA + i) * 2 + (pp >> 1)) ^ sw;
                    va[h][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(sy + row * YB + chunk * 16 + (pp & 1) * 8));
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int chunk = ((wc * SB + j) * 2 + (pp >> 1)) ^ sw;
                    vb[h][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(sx + row * XB + chunk * 16 + (pp & 1) * 8));
                }
            }
            frag_t fa[SA], fb[SB];
#pragma unroll
            for (int i = 0; i < SA; ++i) {
                const s16x8 ta = {va[0][i][0], va[0][i][1], va[0][i][2], va[0][i][3], va[1][i][0], va[1][i][1], va[1][i][2], va[1][i][3]};
                fa[i] = __builtin_bit_cast(frag_t, ta);
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const s16x8 tb = {vb[0][j][0], vb[0][j][1], vb[0][j][2], vb[0][j][3], vb[1][j][0], vb[1][j][1], vb[1][j][2], vb[1][j][3]};
                fb[j] = __builtin_bit_cast(frag_t, tb);
            }
#pragma unroll
            for (int i = 0; i < SA; ++i)
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    acc[i][j] = FragW<TI>::mfma(fa[i], fb[j], acc[i][j]);
                    const int done = (kk * SA + i) * SB + j + 1;
                    if (done % PSTEP == 0 && done / PSTEP <= PPW) {
                        if (more) {
                            WG_PIECE(nst, s + 1, done / PSTEP - 1);
                            if (done / PSTEP == PPW) { WG_ADVANCE(); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tail's zero-fill pieces)
    // partial[split][co][tap][ci]: C/D layout col = lane&15 (ci), row = (lane>>4)*4 + r (co). (Offsets by additions: the 128 stores of
    // the big tile used to cost two 32-bit multiplies each.)
    float* out = a.partial + (long long)split * p.cout * a.taps * p.cin;
    const long long rstride = (long long)a.taps * p.cin;  // one output channel further
    const int co_b = co0 + wr * SA * 16 + (lane >> 4) * 4, ci_b = ci0 + wc * SB * 16 + (lane & 15);
    const long long obase = ((long long)co_b * a.taps + tap) * p.cin + ci_b;
#pragma unroll
    for (int i = 0; i < SA; ++i)
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int ci = ci_b + j * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co_b + i * 16 + r;
                if (co < p.cout && ci < p.cin) out[obase + (i * 16 + r) * rstride + j * 16] = acc[i][j][r];
            }
        }
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s partial[s][i]  (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, long long n, int splits, int accumulate,
                                                           float* __restrict__ dw) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    float4 s = accumulate ? *reinterpret_cast<const float4*>(dw + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = 0; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (long long)k * n + i4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(dw + i4) = s;
}

#ifdef OSR_EXPERIMENT
#include <stdlib.h>
static int wg_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define WG_KNOB(name, dflt) ([] { static const int v = wg_env_int(name, dflt); return v; }())
#else
#define WG_KNOB(name, dflt) (dflt)
#endif
static bool wgrad_big(const osr_conv_params* p) { return p->cout % 256 == 0 && p->cin % 256 == 0; }

static int wgrad_splits(const osr_conv_params* p) {
    const long long M = (long long)p->n * p->ho * p->wo;
    const int tt = wgrad_big(p) ? 256 : 128;
    const long long ntiles = (long long)((p->cout + tt - 1) / tt) * ((p->cin + tt - 1) / tt) * p->kh * p->kw;
    // Workgroups wanted. Every split writes a full fp32 copy of the tile's weights and the reduction reads it back: for the layers
    // with few tiles (the 1x1 layers of res3 / res4: 4 tiles of 256 x 256, K loop of 67 200 rows) that traffic outweighs the operands
    // (128 splits x 1 MB written and re-read against 170 MB of x + dy), and the 256 x 256 tile holds one workgroup per CU anyway, so
    // fewer, longer splits cost no occupancy. Swept inside the training step, where the launches share the GPU with the data-gradient
    // stream (same box, ms per iteration): 512 / 1024 (rounds 1-2) 32.2; 256 / 1024 31.4-31.6; 192 / 1024 30.8; 160 / 1024 30.7;
    // 128 / 1024 31.6; 192 / 384 30.6.
    const long long target = wgrad_big(p) ? WG_KNOB("OSR_WGRAD_TARGET_BIG", 192) : WG_KNOB("OSR_WGRAD_TARGET_SMALL", 384);
    long long splits = (target + ntiles - 1) / ntiles;
    const long long max_splits = (M + 255) / 256;     // at least 256 rows per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return (int)splits;
}

extern "C" int64_t osr_conv2d_wgrad_workspace_bytes(const osr_conv_params* p) {
    if (!p || p->cout < 1 || p->cin < 1 || p->kh < 1 || p->kw < 1 || p->n < 1 || p->ho < 1 || p->wo < 1) {
        osr_set_error("osr_conv2d_wgrad_workspace_bytes: bad parameters");
        return OSR_ERR_INVALID_ARG;
    }
    return (int64_t)wgrad_splits(p) * p->cout * p->kh * p->kw * p->cin * 4;
}

extern "C" osr_status osr_conv2d_wgrad(const osr_conv_params* p, const void* x, const void* dy, float* dw, int32_t accumulate, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(p && x && dy && dw && workspace, OSR_ERR_INVALID_ARG, "osr_conv2d_wgrad: null pointer");
    OSR_REQUIRE(p->n >= 1 && p->hi >= 1 && p->wi >= 1 && p->ho >= 1 && p->wo >= 1, OSR_ERR_INVALID_ARG, "osr_conv2d_wgrad: bad spatial sizes");
    OSR_REQUIRE(p->cin >= 8 && p->cin % 8 == 0 && p->cout >= 8 && p->cout % 8 == 0, OSR_ERR_UNSUPPORTED, "osr_conv2d_wgrad: cin and cout must be multiples of 8");
    OSR_REQUIRE(p->kh >= 1 && p->kw >= 1 && p->kh <= 16 && p->kw <= 16 && p->stride_h >= 1 && p->stride_w >= 1 && p->pad_h >= 0 && p->pad_w >= 0,
                OSR_ERR_INVALID_ARG, "osr_conv2d_wgrad: bad kernel geometry");
    OSR_REQUIRE(p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_conv2d_wgrad: x / dy must be f16 or bf16");
    OSR_REQUIRE(p->pad_mode == 0, OSR_ERR_UNSUPPORTED, "osr_conv2d_wgrad: pad_mode 1 (stem view) has no weight gradient: the stem is frozen");
    OSR_REQUIRE((p->hi + 2 * p->pad_h - p->kh) / p->stride_h + 1 == p->ho && (p->wi + 2 * p->pad_w - p->kw) / p->stride_w + 1 == p->wo, OSR_ERR_INVALID_ARG,
                "osr_conv2d_wgrad: ho/wo inconsistent with hi/wi/kernel/stride/pad");
    OSR_REQUIRE(p->out_stride_w == p->cout && p->out_stride_h == (int64_t)p->wo * p->cout && p->out_stride_n == (int64_t)p->ho * p->wo * p->cout,
                OSR_ERR_UNSUPPORTED, "osr_conv2d_wgrad: dy must be dense (n, ho, wo, cout)");
    OSR_REQUIRE(p->in_stride_w % 8 == 0 && p->in_stride_h % 8 == 0 && p->in_stride_n % 8 == 0 && p->in_stride_n > 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_wgrad: x strides must be multiples of 8 elements");
    OSR_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_conv2d_wgrad: pointers must be 16-byte aligned");
    WgradArgs a;
    a.p = *p; a.x = x; a.dy = dy; a.partial = (float*)workspace;
    a.M = (long long)p->n * p->ho * p->wo;
    const long long x_bytes = (long long)p->n * p->in_stride_n * 2, dy_bytes = a.M * p->cout * 2;
    OSR_REQUIRE(a.M < (1ll << 31) - 1024 && x_bytes < (1ll << 31) - 4096 && dy_bytes < (1ll << 31) - 4096, OSR_ERR_UNSUPPORTED,
                "osr_conv2d_wgrad: tensor too large for 32-bit buffer offsets");
    a.x_bytes = (unsigned)x_bytes; a.dy_bytes = (unsigned)dy_bytes;
    a.splits = wgrad_splits(p);
    a.rows_per_split = ((a.M + a.splits - 1) / a.splits + WG_BM - 1) / WG_BM * WG_BM;
    a.splits = (int)((a.M + a.rows_per_split - 1) / a.rows_per_split);
    const bool big = wgrad_big(p);
    const int tt = big ? 256 : 128;
    a.tiles_co = (p->cout + tt - 1) / tt; a.tiles_ci = (p->cin + tt - 1) / tt; a.taps = p->kh * p->kw;
    a.div_howo = wdiv_make((unsigned)(p->ho * p->wo));
    a.div_wo = wdiv_make((unsigned)p->wo);
    const long long wn = (long long)p->cout * a.taps * p->cin;
    OSR_REQUIRE(workspace_bytes >= (int64_t)a.splits * wn * 4, OSR_ERR_WORKSPACE, "osr_conv2d_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)a.splits * wn * 4);
    const long long grid = (long long)a.tiles_co * a.tiles_ci * a.taps * a.splits;
    OSR_REQUIRE(grid < (1ll << 31), OSR_ERR_UNSUPPORTED, "osr_conv2d_wgrad: grid too large");
    // One split and nothing to add to: the kernel's "partial" IS the result (same layout), written straight into dw -- no copy through the
    // workspace, no reduction launch (FC1: 196 tiles, 51 MB of weights: the reduction was a 0.25 ms copy on the weight-gradient stream).
    const bool direct = a.splits == 1 && !accumulate;
    if (direct) a.partial = dw;
    hipStream_t st = (hipStream_t)stream;
    if (big) {
        const size_t ldsb = (size_t)WG_NST * WG_BM * (256 + 256) * 2;  // 128 KiB
        static osr_dev_mask attr{0};
        osr_once_per_device(attr, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<f16_t, 256, 256, 2, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<bf16_t, 256, 256, 2, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<f16_t, 256, 256, 2, 4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<bf16_t, 256, 256, 2, 4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        // the 8-phase loop pays a longer prologue (two steps staged before the first MFMA): the one-barrier-per-step loop stays for short splits
        const long long steps_per_split = a.rows_per_split / WG_BM;
#ifdef OSR_EXPERIMENT
        const bool use_p8 = wg_env_int("OSR_WGRAD_PH8", 1) != 0 && steps_per_split >= wg_env_int("OSR_WGRAD_PH8_MINSTEPS", WG_P8_MIN_STEPS);  // (scripts/exp_wgrad8.py)
#else
        const bool use_p8 = steps_per_split >= WG_P8_MIN_STEPS;
#endif
        if (!use_p8) {
            if (p->in_dtype == OSR_F16) hipLaunchKernelGGL((conv_wgrad_kernel<f16_t, 256, 256, 2, 4, 0>), dim3((unsigned)grid), dim3(512), ldsb, st, a);
            else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 256, 256, 2, 4, 0>), dim3((unsigned)grid), dim3(512), ldsb, st, a);
        } else if (p->in_dtype == OSR_F16) hipLaunchKernelGGL((conv_wgrad_kernel<f16_t, 256, 256, 2, 4, 1>), dim3((unsigned)grid), dim3(512), ldsb, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 256, 256, 2, 4, 1>), dim3((unsigned)grid), dim3(512), ldsb, st, a);
    } else {
        const size_t ldsb = (size_t)WG_NST * WG_BM * (128 + 128) * 2;  // 64 KiB
        if (p->in_dtype == OSR_F16) hipLaunchKernelGGL((conv_wgrad_kernel<f16_t, 128, 128, 2, 2>), dim3((unsigned)grid), dim3(256), ldsb, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 128, 128, 2, 2>), dim3((unsigned)grid), dim3(256), ldsb, st, a);
    }
    OSR_CHECK_LAUNCH("osr_conv2d_wgrad");
    if (direct) return OSR_OK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((wn / 4 + 255) / 256)), dim3(256), 0, st, (const float*)workspace, wn, a.splits, accumulate, dw);
    OSR_CHECK_LAUNCH("osr_conv2d_wgrad(reduce)");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// bias gradient: db[co] = sum_m dy[m][co]   (two stages, fixed order)
// ------------------------------------------------------------------------------------------------------
// workgroup = 8 row lanes x 32 channel groups of 8 channels (one 16-byte load per thread and row); grid.x row chunks,
// grid.y blocks of 256 channels; the 8 row lanes are summed through LDS in a fixed order
template <class TI>
__global__ __launch_bounds__(256) void bias_grad_kernel(const TI* __restrict__ dy, long long M, int cout, long long rows_per_block, float* __restrict__ partial) {
    __shared__ float s_acc[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int co = blockIdx.y * 256 + cg * 8;
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < cout) {  // cout is a multiple of 8
        long long m = m0 + rl;
        if constexpr (sizeof(TI) == 2) {  // four 16-byte loads in flight per thread (one at a time left the pass at 3.6 TB/s); same order of additions
            typedef TI v8 __attribute__((ext_vector_type(8)));
            for (; m + 24 < m1; m += 32) {
                const v8 v0 = *reinterpret_cast<const v8*>(dy + m * cout + co), v1 = *reinterpret_cast<const v8*>(dy + (m + 8) * cout + co);
                const v8 v2 = *reinterpret_cast<const v8*>(dy + (m + 16) * cout + co), v3 = *reinterpret_cast<const v8*>(dy + (m + 24) * cout + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = (((s[e] + (float)v0[e]) + (float)v1[e]) + (float)v2[e]) + (float)v3[e];
            }
        }
        for (; m < m1; m += 8) {
            if constexpr (sizeof(TI) == 2) {
                typedef TI v8 __attribute__((ext_vector_type(8)));
                const v8 v = *reinterpret_cast<const v8*>(dy + m * cout + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += osr_to_float(dy[m * cout + co + e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_acc[rl][cg * 8 + e] = s[e];
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c < cout) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_acc[r][threadIdx.x];
        partial[(long long)blockIdx.x * cout + c] = t;
    }
}

// any channel count (the 5- and 21-wide fp32 heads): one thread per channel, rows in sequence
template <class TI>
__global__ __launch_bounds__(256) void bias_grad_narrow_kernel(const TI* __restrict__ dy, long long M, int cout, long long rows_per_block, float* __restrict__ partial) {
    const int co = blockIdx.y * 256 + threadIdx.x;
    if (co >= cout) return;
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    float s = 0.f;
    for (long long m = m0; m < m1; ++m) s += osr_to_float(dy[m * cout + co]);
    partial[(long long)blockIdx.x * cout + co] = s;
}

// 8 partial lanes x 32 channels per workgroup; lanes are combined through LDS in a fixed order
__global__ __launch_bounds__(256) void bias_grad_reduce(const float* __restrict__ partial, int nblocks, int cout, int accumulate, float* __restrict__ db) {
    __shared__ float s_acc[8][33];
    const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int co = blockIdx.x * 32 + c;
    float s = 0.f;
    if (co < cout)
        for (int b = rl; b < nblocks; b += 8) s += partial[(long long)b * cout + co];
    s_acc[rl][c] = s;
    __syncthreads();
    if (rl == 0 && co < cout) {
        float t = accumulate ? db[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_acc[r][c];
        db[co] = t;
    }
}

#define BG_BLOCKS 512
extern "C" osr_status osr_bias_grad(const void* dy, int32_t dtype, int64_t m, int32_t cout, float* db, int32_t accumulate, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(dy && db && workspace, OSR_ERR_INVALID_ARG, "osr_bias_grad: null pointer");
    OSR_REQUIRE(m >= 1 && cout >= 1 && osr_dtype_ok(dtype), OSR_ERR_INVALID_ARG, "osr_bias_grad: bad sizes / dtype");
    long long rpb = (m + BG_BLOCKS - 1) / BG_BLOCKS;
    if (rpb < 64) rpb = 64;
    const int nb = (int)((m + rpb - 1) / rpb);
    OSR_REQUIRE(workspace_bytes >= (int64_t)nb * cout * 4, OSR_ERR_WORKSPACE, "osr_bias_grad: workspace needs %lld bytes", (long long)BG_BLOCKS * cout * 4);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(nb, (cout + 255) / 256);
    const bool wide = cout % 8 == 0 && ((uintptr_t)dy & 15) == 0;
#define BG(T) { if (wide) hipLaunchKernelGGL(bias_grad_kernel<T>, grid, dim3(256), 0, st, (const T*)dy, (long long)m, cout, rpb, (float*)workspace); \
                else hipLaunchKernelGGL(bias_grad_narrow_kernel<T>, grid, dim3(256), 0, st, (const T*)dy, (long long)m, cout, rpb, (float*)workspace); }
    if (dtype == OSR_F16) BG(f16_t) else if (dtype == OSR_BF16) BG(bf16_t) else BG(float)
#undef BG
    OSR_CHECK_LAUNCH("osr_bias_grad");
    hipLaunchKernelGGL(bias_grad_reduce, dim3((cout + 31) / 32), dim3(256), 0, st, (const float*)workspace, nb, cout, accumulate, db);
    OSR_CHECK_LAUNCH("osr_bias_grad(reduce)");
    return OSR_OK;
}
