// One whole ResNet bottleneck block of res2 in ONE launch (osr_bottleneck_fwd, include/osr.h):
//     y = relu( conv3_1x1( relu( conv2_3x3( relu( conv1_1x1(x) ) ) ) ) + shortcut(x) ),   cmid = 64, cout = 256, stride 1,
// shortcut = identity (cin = 256: res2.1, res2.2) or a 1x1 projection (cin = 64: res2.0) -- [d2] BottleneckBlock.forward as
// /root/reference/configs/Base-RCNN-FPN.yaml:3-8 (build_resnet_fpn_backbone, STRIDE_IN_1X1, FrozenBN folded) instantiates it.
//
// Why: at 200 x 336 x 16 images the three convolutions of a res2 block are HBM-bound (5-6 TB/s each) and their sum moves
// 2.2 GB; fused, the block reads x once and writes y once (1.1 GB). The two 64-channel intermediates never leave the chip.
//
// One workgroup (4 waves) = one 8 x 16 tile of output pixels of one image:
//   1. conv1 on the 10 x 18 halo of the tile (180 pixels, padded to 192 rows): the pixels are the MFMA's B operand, loaded
//      straight from global memory as fragments (every x value is used by this workgroup once: no LDS staging), the weights
//      (A operand) come from LDS (LDS-DMA, all of w1). The result, relu(. + b1) in the storage dtype, is ZERO outside the image
//      (conv2's zero padding applies to conv1's OUTPUT), and is parked in LDS (mid1).
//   2. conv2: 9 taps x K = 64 from mid1 (shifted row windows of the halo), weights of one tap at a time through a 4-slot
//      LDS-DMA ring (three taps in flight behind a counted vmcnt and a raw barrier). relu(. + b2) -> LDS (mid2).
//   3. conv3 (+ projection): each wave owns 64 of the 256 output channels; its weight fragments come from global memory once
//      and stay in registers; pixels from mid2 (and, for the projection, x's centre pixels from global memory as extra K).
//      bias, identity residual (the same address as the store), ReLU, 16-byte stores.
// What bounds it (round 3, profiles/r03_bottleneck_*.txt): not HBM (1.1-1.7 GB per block at 3.3 TB/s), not the matrix cores (23 %
// busy) -- the per-CU fill rate of the vector memory pipe. A tile moves 360 KB through it (halo pixels 96 KB, residual 64 KB,
// weights 136 KB, stores 64 KB); at the 23-70 GB/s a CU draws from HBM / Infinity Cache / L2 (MI355X_MICROARCH.md, indexed rows)
// that is ~10 us per tile and CU = 335 us per block, which is what it takes. Re-arrangements that only move waiting around
// measured the same or worse (all loads of conv3 requested under conv2; a channel-split barrier-free conv2: twice the LDS
// reads; persistent workgroups that stage the next tile's w1 and request its pixels under conv3: entry 6.9 -> 1.4 us per tile,
// conv2 / conv3 longer by as much). What would help is fewer bytes per pixel: 16 x 16 tiles (weights per pixel halved, halo 1.27x
// instead of 1.41x) and the residual taken from conv1's fragments instead of a second read.
// The MFMA operand roles are SWAPPED against osr_conv_gemm64.hip (weights = A, pixels = B): D = W . X^T puts four consecutive
// output channels of one pixel into each lane, and with the row permutation below two MFMAs give a lane eight consecutive
// channels = one 16-byte chunk -- LDS writes (ds_write_b128) and global stores need no transposition through LDS.
// The K order (k ascending inside a 1x1, tap-major for the 3x3 with 64 channels per tap) and the rounding points (fp16/bf16
// after every ReLU) are those of three osr_conv2d_fwd launches; the projection block differs in ONE rounding (the shortcut's
// output is never rounded to the storage dtype: it is accumulated in fp32 with conv3).
#include "osr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f16_t bnf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bnbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned bnu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void bn_lds_void_t;

template <class T> struct BnFrag;
template <> struct BnFrag<f16_t> {
    typedef bnf16x8 type;
    static __device__ __forceinline__ f32x4 mfma(bnf16x8 a, bnf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct BnFrag<bf16_t> {
    typedef bnbf16x8 type;
    static __device__ __forceinline__ f32x4 mfma(bnbf16x8 a, bnbf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

struct BnArgs {
    const void* x;
    void* y;
    const void *w1, *w2, *w3, *wsc;     // packed [cout][kh][kw][cin], storage dtype
    const float *b1, *b2, *b3, *bsc;    // fp32
    int n, h, w;
    int tiles_y, tiles_x;
    unsigned x_bytes, y_bytes, w1_bytes, w2_bytes, w3_bytes, wsc_bytes;
#ifdef BN_STAMPS
    unsigned long long* dbg;
#endif
};

// Diagnostic build only (-DBN_STAMPS, never shipped; scripts/exp_bottleneck_stamps.py): wave 0 of every workgroup records
// s_memrealtime (100 MHz) at entry, when conv1 may start, after conv1, after conv2, after conv3 and when its stores have drained.
#ifdef BN_STAMPS
static unsigned long long* g_bn_stamps = nullptr;
extern "C" void osr_debug_set_bn_stamps(unsigned long long* p) { g_bn_stamps = p; }
#define BN_STAMP(i) if (a.dbg && tid == 0) a.dbg[(long long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define BN_STAMP(i)
#endif

#define BN_TH 8
#define BN_TW 16
#define BN_HW 18                 // halo width
#define BN_NHALO 180             // 10 x 18 halo pixels
#define BN_MID1 0                // 192 rows x 128 B
#define BN_MID2 24576            // 128 rows x 128 B
#define BN_WBUF 40960            // 32 KB: all of w1 (cin = 256: four K slices of 8 KB), then the ring of four conv2 taps
#define BN_BIAS 73728            // b1 (64), b2 (64), b3 [+ bsc] (256) fp32
#define BN_LDS (73728 + 384 * 4)
#define BN_OOB 0x80000000u

// Row n of a 64-row weight block sits at LDS row rho(n): the 16 rows an MFMA's A operand takes are consecutive in LDS (so the
// fragment read is conflict-free under the usual XOR swizzle), and they are the channels
//     n(T, i) = 32 (T >> 1) + 8 (i >> 2) + 4 (T & 1) + (i & 3),     T = 16-row tile 0..3, i = row in the tile,
// so that the lane that holds rows 4g..4g+3 of the output tiles 2p and 2p+1 holds channels 32p + 8g .. + 7: one 16-byte chunk.
__device__ __forceinline__ int bn_nmap(int rho) {
    const int T = rho >> 4, i = rho & 15;
    return 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);
}

template <class TI> __device__ __forceinline__ typename BnFrag<TI>::type bn_as_frag(bnu32x4 v) {
    union { bnu32x4 u; typename BnFrag<TI>::type f; } c;
    c.u = v;
    return c.f;
}

template <class TI, int CIN, int PROJ>
__global__ __launch_bounds__(256, 2) void bottleneck64_kernel(BnArgs a) {
    typedef typename BnFrag<TI>::type frag_t;
    constexpr int NK32 = CIN / 32, NKS = CIN / 64;
    static_assert(CIN == 64 || CIN == 256, "res2 blocks only");
    static_assert(PROJ ? CIN == 64 : CIN == 256, "identity shortcut needs cin == cout");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    BN_STAMP(0);

    // XCD-aware bijective remap of the linear block id: an XCD walks a contiguous run of tiles (row-major inside an image), so the
    // halo pixels a tile shares with its neighbours are fetched into one L2
    int t;
    {
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, img = t / (a.tiles_x * a.tiles_y);
    const int y0 = ty * BN_TH, x0 = tx * BN_TW;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w1), 0, a.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w2), 0, a.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w3), 0, a.w3_bytes, 0x00020000);

    // ---- 0. weights of conv1 -> LDS (LDS-DMA: piece = 8 rows x 128 B; lane = row piece*8 + lane/8, slot lane%8 of the LDS row,
    //         which receives logical chunk slot ^ ((rho >> 1) & 7) of weight row nmap(rho)); biases -> LDS ----
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = wid * 2 + j;
                const int rho = piece * 8 + lrow;
                const unsigned chunk = (unsigned)(slot ^ ((rho >> 1) & 7));
                const unsigned voff = (unsigned)((bn_nmap(rho) * CIN + ks * 64 + (int)chunk * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (bn_lds_void_t*)(lds + BN_WBUF + ks * 8192 + piece * 1024), 16, voff, 0, 0, 0);
            }
        float* s_bias = reinterpret_cast<float*>(lds + BN_BIAS);
        for (int i = tid; i < 384; i += 256) {
            float v;
            if (i < 64) v = a.b1[i];
            else if (i < 128) v = a.b2[i - 64];
            else v = a.b3[i - 128] + (PROJ ? a.bsc[i - 128] : 0.f);
            s_bias[i] = v;
        }
    }

    // ---- 1. conv1 on the halo: this wave's 48 halo rows = 3 pixel tiles of 16; pixel fragments straight from global memory ----
    unsigned xoff[3];
    bool hvalid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int hr = wid * 48 + i * 16 + l15;
        const int hy = hr / BN_HW, hx = hr - hy * BN_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        hvalid[i] = hr < BN_NHALO && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        xoff[i] = hvalid[i] ? (unsigned)((((long long)img * a.h + gy) * a.w + gx) * CIN * 2 + g * 16) : BN_OOB;
    }
    frag_t xf[3][NK32];
#pragma unroll
    for (int k = 0; k < NK32; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) xf[i][k] = bn_as_frag<TI>(__builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[i] + k * 64, 0, 0));

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (w1's LDS-DMA pieces of this wave; the pixel loads with them)
    __syncthreads();
    BN_STAMP(1);

    // swizzled byte offset of this lane's 16-byte read of LDS row `row`, logical chunk `chunk` (rows of 128 B)
#define BN_SW(row, chunk) ((row) * 128 + ((((chunk)) ^ (((row) >> 1) & 7)) << 4))
    {
        f32x4 acc[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[i][T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NK32; ++k) {
            frag_t wf[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const int rho = T * 16 + l15;
                wf[T] = *reinterpret_cast<const frag_t*>(lds + BN_WBUF + (k >> 1) * 8192 + BN_SW(rho, (k & 1) * 4 + g));
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int T = 0; T < 4; ++T) acc[i][T] = BnFrag<TI>::mfma(wf[T], xf[i][k], acc[i][T]);
        }
        const float* s_b1 = reinterpret_cast<const float*>(lds + BN_BIAS);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hr = wid * 48 + i * 16 + l15;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 ba = *reinterpret_cast<const float4*>(s_b1 + 32 * p + 8 * g), bb = *reinterpret_cast<const float4*>(s_b1 + 32 * p + 8 * g + 4);
                const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
                frag_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (e < 4 ? acc[i][2 * p][e] : acc[i][2 * p + 1][e - 4]) + bias[e];
                    o[e] = (TI)(hvalid[i] ? fmaxf(v, 0.f) : 0.f);  // outside the image conv2 sees ZEROS, not relu(b1)
                }
                *reinterpret_cast<frag_t*>(lds + BN_MID1 + BN_SW(hr, 4 * p + g)) = o;
            }
        }
    }
    __syncthreads();  // mid1 complete; every wave is done with w1
    BN_STAMP(2);

    // ---- conv3's weight fragments: global -> registers, in flight under conv2 (this wave's 64 output channels) ----
    frag_t w3f[4][2];
    frag_t wscf[PROJ ? 4 : 1][2];
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = wid * 64 + bn_nmap(T * 16 + l15);
            w3f[T][k] = bn_as_frag<TI>(__builtin_amdgcn_raw_buffer_load_b128(rs_w3, (unsigned)((n * 64 + k * 32 + g * 8) * 2), 0, 0));
            if constexpr (PROJ) {
                const __amdgpu_buffer_rsrc_t rs_wsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wsc), 0, a.wsc_bytes, 0x00020000);
                wscf[T][k] = bn_as_frag<TI>(__builtin_amdgcn_raw_buffer_load_b128(rs_wsc, (unsigned)((n * CIN + k * 32 + g * 8) * 2), 0, 0));
            }
        }

    // ---- 2. conv2: tap ring (slot = tap & 3), three taps in flight ----
#define BN_STAGE_TAP(tap_)                                                                                                          \
    {                                                                                                                               \
        const int lrow_ = lane >> 3, slot_ = lane & 7;                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                                          \
            const int piece_ = wid * 2 + j_;                                                                                        \
            const int rho_ = piece_ * 8 + lrow_;                                                                                    \
            const unsigned chunk_ = (unsigned)(slot_ ^ ((rho_ >> 1) & 7));                                                          \
            const unsigned voff_ = (unsigned)(((bn_nmap(rho_) * 9 + (tap_)) * 64 + (int)chunk_ * 8) * 2);                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (bn_lds_void_t*)(lds + BN_WBUF + ((tap_) & 3) * 8192 + piece_ * 1024), 16, voff_, 0, 0, 0); \
        }                                                                                                                           \
    }
    BN_STAGE_TAP(0);
    BN_STAGE_TAP(1);
    BN_STAGE_TAP(2);
    {
        f32x4 acc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[j][T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // RAW: tap `tap` has landed for this wave when at most the pieces of the taps issued after it are outstanding (counted
            // vmcnt), and for every wave behind the barrier. WAR: the slot restaged below, (tap + 3) & 3, is the one tap - 1 was read
            // from -- the compiler rotates the loop, so a wave's last fragment reads of tap - 1 are issued just in front of this
            // barrier and may still be IN FLIGHT when it arrives: lgkmcnt(0) retires them first. Without it another wave's LDS-DMA
            // could land in the slot before a queued ds_read had been served -- seen as one wrong tile in a few thousand, and only in
            // the first launches after the GPU had idled (at a low shader clock the DMA's latency shrinks in shader cycles).
            // Wait and barrier are ONE asm statement: s_barrier alone is no compiler barrier for memory operations.
            if (tap <= 6) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (tap == 7) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tap + 3 < 9) BN_STAGE_TAP(tap + 3);
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned char* wb = lds + BN_WBUF + (tap & 3) * 8192;
            frag_t wf[4][2];
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int k = 0; k < 2; ++k) wf[T][k] = *reinterpret_cast<const frag_t*>(wb + BN_SW(T * 16 + l15, k * 4 + g));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hr = (wid * 2 + j + ky) * BN_HW + kx + l15;  // output row oy = 2 wid + j, pixel ox = l15, shifted by the tap
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const frag_t af = *reinterpret_cast<const frag_t*>(lds + BN_MID1 + BN_SW(hr, k * 4 + g));
#pragma unroll
                    for (int T = 0; T < 4; ++T) acc[j][T] = BnFrag<TI>::mfma(wf[T][k], af, acc[j][T]);
                }
            }
        }
        const float* s_b2 = reinterpret_cast<const float*>(lds + BN_BIAS) + 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wid * 2 + j) * 16 + l15;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 ba = *reinterpret_cast<const float4*>(s_b2 + 32 * p + 8 * g), bb = *reinterpret_cast<const float4*>(s_b2 + 32 * p + 8 * g + 4);
                const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
                frag_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (TI)fmaxf((e < 4 ? acc[j][2 * p][e] : acc[j][2 * p + 1][e - 4]) + bias[e], 0.f);
                *reinterpret_cast<frag_t*>(lds + BN_MID2 + BN_SW(row, 4 * p + g)) = o;
            }
        }
    }
    __syncthreads();  // mid2 complete
    BN_STAMP(3);

    // ---- 3. conv3 (+ projection) for this wave's 64 channels, 32 pixels (two output rows) at a time ----
    const float* s_b3 = reinterpret_cast<const float*>(lds + BN_BIAS) + 128 + wid * 64;
    float bias3[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float4 ba = *reinterpret_cast<const float4*>(s_b3 + 32 * p + 8 * g), bb = *reinterpret_cast<const float4*>(s_b3 + 32 * p + 8 * g + 4);
        bias3[p][0] = ba.x; bias3[p][1] = ba.y; bias3[p][2] = ba.z; bias3[p][3] = ba.w;
        bias3[p][4] = bb.x; bias3[p][5] = bb.y; bias3[p][6] = bb.z; bias3[p][7] = bb.w;
    }
    // per group gi: pixel (y0 + 2 gi + j, x0 + l15); `aux` = the identity residual's chunks (one per channel pair p) or, for the
    // projection, x's centre-pixel fragments (one per 32-wide K step)
    frag_t aux[2][2][2];  // [buffer][j][p or k]
    unsigned ooff[2][2];  // [buffer][j]: byte offset of channel 0 of the pixel in y (BN_OOB: outside the image)
#define BN_FETCH(buf, gi_)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                                          \
            const int gy_ = y0 + 2 * (gi_) + j_, gx_ = x0 + l15;                                                                    \
            const bool ok_ = gy_ < a.h && gx_ < a.w;                                                                                \
            const long long pix_ = ((long long)img * a.h + gy_) * a.w + gx_;                                                        \
            ooff[buf][j_] = ok_ ? (unsigned)(pix_ * 256 * 2) : BN_OOB;                                                              \
            _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                                      \
                const unsigned off_ = !ok_ ? BN_OOB : PROJ ? (unsigned)(pix_ * CIN * 2 + (q_ * 32 + g * 8) * 2)                      \
                                                           : (unsigned)(pix_ * 256 * 2 + (wid * 64 + 32 * q_ + 8 * g) * 2);          \
                aux[buf][j_][q_] = bn_as_frag<TI>(__builtin_amdgcn_raw_buffer_load_b128(rs_x, off_, 0, 0));                          \
            }                                                                                                                       \
        }                                                                                                                           \
    }
    BN_FETCH(0, 0);
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        const int cur = gi & 1;
        if (gi + 1 < 4) { BN_FETCH((gi + 1) & 1, gi + 1); }
        f32x4 acc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[j][T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (2 * gi + j) * 16 + l15;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const frag_t af = *reinterpret_cast<const frag_t*>(lds + BN_MID2 + BN_SW(row, k * 4 + g));
#pragma unroll
                for (int T = 0; T < 4; ++T) acc[j][T] = BnFrag<TI>::mfma(w3f[T][k], af, acc[j][T]);
            }
            if constexpr (PROJ) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int T = 0; T < 4; ++T) acc[j][T] = BnFrag<TI>::mfma(wscf[T][k], aux[cur][j][k], acc[j][T]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                frag_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (e < 4 ? acc[j][2 * p][e] : acc[j][2 * p + 1][e - 4]) + bias3[p][e];
                    if constexpr (!PROJ) v += (float)aux[cur][j][p][e];
                    o[e] = (TI)fmaxf(v, 0.f);
                }
                union { frag_t f; bnu32x4 u; } cv;
                cv.f = o;
                // (an out-of-image pixel's offset is out of the buffer's range: the store is dropped by the bounds check)
                __builtin_amdgcn_raw_buffer_store_b128(cv.u, rs_y, ooff[cur][j] == BN_OOB ? BN_OOB : ooff[cur][j] + (unsigned)((wid * 64 + 32 * p + 8 * g) * 2), 0, 0);
            }
    }
    BN_STAMP(4);
#ifdef BN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (stamp 5: this wave's stores have left the CU)
    BN_STAMP(5);
#endif
}

template <class TI>
static osr_status bottleneck_launch(const osr_bottleneck_params* p, BnArgs& a, hipStream_t st) {
    const unsigned grid = (unsigned)(a.n * a.tiles_y * a.tiles_x);
    if (p->has_proj) {
        static osr_dev_mask m{0};
        osr_once_per_device(m, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck64_kernel<TI, 64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS); });
        hipLaunchKernelGGL((bottleneck64_kernel<TI, 64, 1>), dim3(grid), dim3(256), BN_LDS, st, a);
    } else {
        static osr_dev_mask m{0};
        osr_once_per_device(m, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck64_kernel<TI, 256, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS); });
        hipLaunchKernelGGL((bottleneck64_kernel<TI, 256, 0>), dim3(grid), dim3(256), BN_LDS, st, a);
    }
    OSR_CHECK_LAUNCH("osr_bottleneck_fwd");
    return OSR_OK;
}

extern "C" osr_status osr_bottleneck_fwd(const osr_bottleneck_params* p, const void* in, const void* w1, const float* b1, const void* w2,
                                         const float* b2, const void* w3, const float* b3, const void* wsc, const float* bsc, void* out,
                                         void* stream) {
    OSR_REQUIRE(p && in && w1 && b1 && w2 && b2 && w3 && b3 && out, OSR_ERR_INVALID_ARG, "osr_bottleneck_fwd: null pointer");
    OSR_REQUIRE(p->n >= 1 && p->h >= 1 && p->w >= 1, OSR_ERR_INVALID_ARG, "osr_bottleneck_fwd: bad geometry");
    OSR_REQUIRE(p->dtype == OSR_F16 || p->dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_bottleneck_fwd: dtype must be f16/bf16");
    OSR_REQUIRE(p->cmid == 64 && p->cout == 256 && ((p->has_proj && p->cin == 64) || (!p->has_proj && p->cin == 256)), OSR_ERR_UNSUPPORTED,
                "osr_bottleneck_fwd: only the res2 shapes are fused (cmid 64, cout 256; cin 64 with a projection or cin 256 without): run "
                "three osr_conv2d_fwd launches instead");
    OSR_REQUIRE(!p->has_proj || (wsc && bsc), OSR_ERR_INVALID_ARG, "osr_bottleneck_fwd: the projection shortcut needs its weight and bias");
    OSR_REQUIRE(in != out, OSR_ERR_INVALID_ARG, "osr_bottleneck_fwd: in-place is not supported (neighbouring tiles read the halo)");
    OSR_REQUIRE((((uintptr_t)in | (uintptr_t)out | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)wsc | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)b3) & 15) == 0,
                OSR_ERR_INVALID_ARG, "osr_bottleneck_fwd: pointers must be 16-byte aligned");
    const long long px = (long long)p->n * p->h * p->w;
    const long long xb = px * p->cin * 2, yb = px * p->cout * 2;
    OSR_REQUIRE(xb < (1ll << 31) - 4096 && yb < (1ll << 31) - 4096, OSR_ERR_UNSUPPORTED, "osr_bottleneck_fwd: tensor too large for 32-bit buffer offsets");
    BnArgs a;
    a.x = in; a.y = out; a.w1 = w1; a.w2 = w2; a.w3 = w3; a.wsc = wsc; a.b1 = b1; a.b2 = b2; a.b3 = b3; a.bsc = bsc;
    a.n = p->n; a.h = p->h; a.w = p->w;
    a.tiles_y = (p->h + BN_TH - 1) / BN_TH; a.tiles_x = (p->w + BN_TW - 1) / BN_TW;
#ifdef BN_STAMPS
    a.dbg = g_bn_stamps;
#endif
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
    a.w1_bytes = (unsigned)(64 * p->cin * 2); a.w2_bytes = 64 * 9 * 64 * 2; a.w3_bytes = 256 * 64 * 2; a.wsc_bytes = (unsigned)(256 * p->cin * 2);
    OSR_REQUIRE((long long)a.n * a.tiles_y * a.tiles_x < (1ll << 31), OSR_ERR_UNSUPPORTED, "osr_bottleneck_fwd: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    return p->dtype == OSR_F16 ? bottleneck_launch<f16_t>(p, a, st) : bottleneck_launch<bf16_t>(p, a, st);
}
