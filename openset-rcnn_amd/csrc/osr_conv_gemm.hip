// Implicit-GEMM convolution / fully-connected layers on the CDNA4 matrix cores (include/osr.h: osr_conv2d_fwd),
// and the exact-fp32 MFMA GEMM of the PLN head (osr_gemm_f32).
//
// Replaces [d2] Conv2d(+FrozenBatchNorm2d folded)+ReLU(+shortcut) of build_resnet_fpn_backbone
// (/root/reference/configs/Base-RCNN-FPN.yaml:3-8), the FPN lateral/output convs with the nearest-2x top-down add,
// ClsFreeRPNHead.conv (classification_free_rpn.py:158) and FastRCNNConvFCHead fc1/fc2 (osrcnn_roi_heads.py:308).
//
// Layout: activations NHWC, weights [cout][kh][kw][cin] (K contiguous), so a BK=32 slice of the GEMM K axis is 64
// contiguous bytes of one input pixel: the A-operand gather is one 16-byte load per lane. GEMM view:
//   M = n*ho*wo output pixels (rows), N = cout, K = kh*kw*cin.
// Tile: BM x BN x 32 per 256-thread workgroup (4 waves), v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate,
// LDS double buffer with 80-byte rows (conflict-free ds_read_b128 of the 32x32x16 fragments), register-staged
// prefetch of the next K slice behind the current slice's MFMAs, one barrier per K step. The epilogue goes
// through a wave-private LDS slab so that bias + residual (+ FPN upsample-add) + ReLU + down-convert are applied
// on 16-byte row segments and the store is coalesced along channels.
#include "osr_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class T> struct Frag;
template <> struct Frag<f16_t> {
    typedef f16x8 type;
    static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Frag<bf16_t> {
    typedef bf16x8 type;
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

#define CV_BK 32
#define CV_ROWB 80  // LDS bytes per tile row: 64 data + 16 pad

struct ConvArgs {
    osr_conv_params p;
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    void* out;
    long long M;   // n*ho*wo
    int K;         // kh*kw*cin
    int tiles_m, tiles_n;
};

template <class TO> __device__ __forceinline__ void store8(TO* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float v[8]) {
    f16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (f16_t)v[i];
    *reinterpret_cast<f16x8*>(p) = t;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float v[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}

template <class TI, class TO, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int A_CH = BM * 4 / 256, B_CH = BN * 4 / 256;  // 16-byte chunks per thread per K step
    constexpr int STAGE = (BM + BN) * CV_ROWB;
    constexpr int EPI_LD = TN * 32 + 4;                       // floats per staged row
    constexpr int EPI_BYTES = 4 * 32 * EPI_LD * 4;
    constexpr int LDS_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    typedef typename Frag<TI>::type frag_t;
    const osr_conv_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const int tile_n = blockIdx.x % a.tiles_n, tile_m = blockIdx.x / a.tiles_n;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const TI* __restrict__ in = reinterpret_cast<const TI*>(a.in);
    const TI* __restrict__ wgt = reinterpret_cast<const TI*>(a.w);
    const int howo = p.ho * p.wo;

    // ---- per-thread gather descriptors (fixed over the K loop) ----
    long long a_base[A_CH];
    int a_ih0[A_CH], a_iw0[A_CH];
    bool a_ok[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int q = tid + 256 * i, row = q >> 2;
        const long long m = m0 + row;
        a_ok[i] = m < a.M;
        const long long mm = a_ok[i] ? m : 0;
        const int nimg = (int)(mm / howo), rem = (int)(mm - (long long)nimg * howo);
        const int oh = rem / p.wo, ow = rem - oh * p.wo;
        a_ih0[i] = oh * p.stride_h - p.pad_h;
        a_iw0[i] = ow * p.stride_w - p.pad_w;
        a_base[i] = (long long)nimg * p.in_stride_n + (q & 3) * 8;
    }
    long long b_off[B_CH];
    bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int q = tid + 256 * i, row = q >> 2;
        b_ok[i] = n0 + row < p.cout;
        b_off[i] = (long long)(b_ok[i] ? n0 + row : 0) * a.K + (q & 3) * 8;
    }

    u32x4 ra[A_CH], rb[B_CH];
    int kh = 0, kw = 0, c0 = 0, kflat = 0;

    // (macros, not lambdas: by-reference captures of the register arrays would force them into scratch)
#define CV_LOAD_TILES()                                                                                                      \
    {                                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                                                   \
            const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;                                                                \
            bool ok = a_ok[i];                                                                                               \
            if (p.pad_mode == 0) ok = ok && (unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi;                  \
            const long long off = ok ? a_base[i] + (long long)ih * p.in_stride_h + (long long)iw * p.in_stride_w + c0 : 0ll; \
            const u32x4 v = *reinterpret_cast<const u32x4*>(in + off);                                                       \
            const unsigned int msk = ok ? 0xffffffffu : 0u;                                                                  \
            ra[i] = v & msk;                                                                                                 \
        }                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                                                                   \
            const u32x4 v = *reinterpret_cast<const u32x4*>(wgt + b_off[i] + kflat);                                         \
            const unsigned int msk = b_ok[i] ? 0xffffffffu : 0u;                                                             \
            rb[i] = v & msk;                                                                                                 \
        }                                                                                                                    \
    }
#define CV_ADVANCE()                                         \
    {                                                        \
        kflat += CV_BK;                                      \
        c0 += CV_BK;                                         \
        if (c0 >= p.cin) { c0 = 0; if (++kw >= p.kw) { kw = 0; ++kh; } } \
    }
#define CV_STORE_TILES(buf)                                                                        \
    {                                                                                              \
        unsigned char* sa_ = lds + (buf) * STAGE;                                                  \
        unsigned char* sb_ = sa_ + BM * CV_ROWB;                                                   \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                         \
            const int q = tid + 256 * i;                                                           \
            *reinterpret_cast<u32x4*>(sa_ + (q >> 2) * CV_ROWB + (q & 3) * 16) = ra[i];            \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                                         \
            const int q = tid + 256 * i;                                                           \
            *reinterpret_cast<u32x4*>(sb_ + (q >> 2) * CV_ROWB + (q & 3) * 16) = rb[i];            \
        }                                                                                          \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.K / CV_BK;
    CV_LOAD_TILES();
    CV_STORE_TILES(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) { CV_ADVANCE(); CV_LOAD_TILES(); }
        const unsigned char* sa = lds + (ks & 1) * STAGE;
        const unsigned char* sb = sa + BM * CV_ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const frag_t*>(sa + ((wr * TM + i) * 32 + (lane & 31)) * CV_ROWB + kk * 32 + (lane >> 5) * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const frag_t*>(sb + ((wc * TN + j) * 32 + (lane & 31)) * CV_ROWB + kk * 32 + (lane >> 5) * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Frag<TI>::mfma(fa[i], fb[j], acc[i][j]);
        }
        if (more) CV_STORE_TILES((ks + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: acc -> wave-private LDS slab (32 rows x TN*32 fp32) -> 8 channels per lane ----
    float* slab = reinterpret_cast<float*>(lds) + wid * 32 * EPI_LD;
    TO* __restrict__ out = reinterpret_cast<TO*>(a.out);
    const TI* __restrict__ res = reinterpret_cast<const TI*>(a.res);
    constexpr int LPR = TN * 4;        // lanes per staged row (8 channels each)
    constexpr int RPP = 64 / LPR;      // rows per pass
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                slab[row * EPI_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): wave-private slab, no barrier needed
        __builtin_amdgcn_wave_barrier();
        const int cseg = (lane % LPR) * 8;
        const int co = n0 + wc * TN * 32 + cseg;
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
            const int row = pass * RPP + lane / LPR;
            const long long m = m0 + (wr * TM + i) * 32 + row;
            if (m < a.M && co < p.cout) {
                const float4 v0 = *reinterpret_cast<const float4*>(slab + row * EPI_LD + cseg);
                const float4 v1 = *reinterpret_cast<const float4*>(slab + row * EPI_LD + cseg + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
                const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                const int nimg = (int)(m / howo), rem = (int)(m - (long long)nimg * howo);
                const int oh = rem / p.wo, ow = rem - oh * p.wo;
                if (p.res_mode != 0) {
                    const int rh = p.res_mode == 2 ? (oh >> 1) : oh, rw = p.res_mode == 2 ? (ow >> 1) : ow;
                    const frag_t rv = *reinterpret_cast<const frag_t*>(res + (long long)nimg * p.res_stride_n + (long long)rh * p.res_stride_h +
                                                                        (long long)rw * p.res_stride_w + co);
                    if (p.res_mode == 3) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)rv[e] > 0.f ? v[e] : 0.f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                store8<TO>(out + (long long)nimg * p.out_stride_n + (long long)oh * p.out_stride_h + (long long)ow * p.out_stride_w + co, v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <class TI, class TO>
static osr_status conv_launch(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    if (a.p.cout <= 64) {
        a.tiles_m = (int)((a.M + 127) / 128);
        a.tiles_n = (a.p.cout + 63) / 64;
        hipLaunchKernelGGL((conv_igemm_kernel<TI, TO, 128, 64, 4, 1>), dim3((unsigned)a.tiles_m * a.tiles_n), dim3(256), 0, st, a);
    } else {
        a.tiles_m = (int)((a.M + 127) / 128);
        a.tiles_n = (a.p.cout + 127) / 128;
        hipLaunchKernelGGL((conv_igemm_kernel<TI, TO, 128, 128, 2, 2>), dim3((unsigned)a.tiles_m * a.tiles_n), dim3(256), 0, st, a);
    }
    OSR_CHECK_LAUNCH("osr_conv2d_fwd");
    return OSR_OK;
}

osr_status osr_conv_f32_run(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual, void* out,
                            hipStream_t st);
int osr_conv64_eligible(const osr_conv_params* p, long long in_bytes, long long w_bytes);
long long osr_conv64_split_workspace_bytes(const osr_conv_params* p);
int osr_conv64_describe(const osr_conv_params* p, int has_workspace, char* buf, int n);
osr_status osr_conv64_run(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual, const void* mask,
                          void* out, long long in_bytes, long long w_bytes, hipStream_t st);

static bool force_bk32() {
#ifdef OSR_EXPERIMENT  // diagnostic builds only: route every layer through the BK=32 kernel
    static const bool v = [] { const char* e = getenv("OSR_CONV_BK32"); return e && e[0] == '1'; }();
    return v;
#else
    return false;
#endif
}

static osr_status conv2d_fwd_impl(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual,
                                  const void* mask, bool masked, void* out, void* stream) {
    OSR_REQUIRE(p && in && weight && bias && out, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: null pointer");
    if (masked) {
        OSR_REQUIRE(mask && (((uintptr_t)mask) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd_masked: mask must be a 16-byte aligned pointer");
        OSR_REQUIRE(p->res_mode == 0 || p->res_mode == 1, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd_masked: res_mode must be 0 or 1");
    }
    OSR_REQUIRE(p->n >= 1 && p->hi >= 1 && p->wi >= 1 && p->ho >= 1 && p->wo >= 1, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: bad spatial sizes");
    OSR_REQUIRE(!p->row_seg_counts || p->row_seg_rows >= 1, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: row_seg_rows must be positive with row_seg_counts");
    const bool f32_mode = p->in_dtype == OSR_F32;  // parity mode: fp32 storage and products (osr_conv_f32.hip)
    OSR_REQUIRE(f32_mode || (p->cin >= 32 && p->cin % 32 == 0), OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd: cin must be a multiple of 32, got %d", p->cin);
    OSR_REQUIRE(f32_mode || (p->cout >= 8 && p->cout % 8 == 0), OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd: cout must be a multiple of 8, got %d", p->cout);
    OSR_REQUIRE(p->cout >= 1, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: cout must be positive");
    OSR_REQUIRE(p->kh >= 1 && p->kw >= 1 && p->kh <= 16 && p->kw <= 16 && p->stride_h >= 1 && p->stride_w >= 1 && p->pad_h >= 0 && p->pad_w >= 0,
                OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: bad kernel geometry");
    OSR_REQUIRE(osr_dtype_ok(p->in_dtype), OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd: in_dtype must be f16/bf16 (fast path) or f32 (parity mode)");
    OSR_REQUIRE(osr_dtype_ok(p->out_dtype), OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: bad out_dtype");
    OSR_REQUIRE(!(f32_mode && masked), OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd_masked: f16/bf16 only");
    OSR_REQUIRE(p->res_mode >= 0 && p->res_mode <= 3 && (p->res_mode == 0 || residual), OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: bad res_mode / residual");
    OSR_REQUIRE(p->pad_mode == 0 || p->pad_mode == 1, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: bad pad_mode");
    if (p->pad_mode == 0) {
        // the computed output size must agree with the convolution arithmetic, so that every tap index the kernel
        // forms is either inside [0,hi)x[0,wi) or rejected by the bounds check
        OSR_REQUIRE((p->hi + 2 * p->pad_h - p->kh) / p->stride_h + 1 == p->ho && (p->wi + 2 * p->pad_w - p->kw) / p->stride_w + 1 == p->wo,
                    OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: ho/wo inconsistent with hi/wi/kernel/stride/pad");
    }
    if (f32_mode) {
        OSR_REQUIRE((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0, OSR_ERR_INVALID_ARG,
                    "osr_conv2d_fwd: pointers must be 16-byte aligned");
        return osr_conv_f32_run(p, in, weight, bias, residual, out, (hipStream_t)stream);
    }
    OSR_REQUIRE(p->in_stride_w % 4 == 0 && p->in_stride_h % 8 == 0 && p->in_stride_n % 8 == 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd: input strides must keep 16-byte alignment (w %% 4, h/n %% 8; w %% 8 unless the stem view)");
    OSR_REQUIRE(p->in_stride_w % 8 == 0 || (p->pad_mode == 1 && (p->in_stride_w * p->stride_w) % 8 == 0), OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd: in_stride_w*stride_w must be a multiple of 8 elements");
    OSR_REQUIRE(p->out_stride_w % 8 == 0 && p->out_stride_h % 8 == 0 && p->out_stride_n % 8 == 0, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: output strides must be multiples of 8");
    if (p->res_mode)
        OSR_REQUIRE(p->res_stride_w % 8 == 0 && p->res_stride_h % 8 == 0 && p->res_stride_n % 8 == 0, OSR_ERR_INVALID_ARG, "osr_conv2d_fwd: residual strides must be multiples of 8");
    OSR_REQUIRE((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd: pointers must be 16-byte aligned");
    ConvArgs a;
    a.p = *p; a.in = in; a.w = weight; a.bias = bias; a.res = residual; a.out = out;
    a.M = (long long)p->n * p->ho * p->wo;
    const long long K = (long long)p->kh * p->kw * p->cin;
    OSR_REQUIRE(K <= (1ll << 30) && a.M <= (1ll << 31) - 256, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd: problem too large");
    a.K = (int)K;
    a.tiles_m = a.tiles_n = 0;
    hipStream_t st = (hipStream_t)stream;
    {
        // extent of the input buffer implied by the strides (exact for contiguous NHWC, the stem view and FC rows)
        const long long in_elems = p->in_stride_n > 0 ? (long long)p->n * p->in_stride_n
                                                      : (long long)(p->hi - 1) * p->in_stride_h + (long long)(p->wi - 1) * p->in_stride_w + p->cin;
        const long long in_bytes = in_elems * 2, w_bytes = (long long)p->cout * K * 2;
        if (!force_bk32() && osr_conv64_eligible(p, in_bytes, w_bytes))
            return osr_conv64_run(p, in, weight, bias, residual, masked ? mask : nullptr, out, in_bytes, w_bytes, st);
    }
    if (masked) {
        osr_set_error("osr_conv2d_fwd_masked: outside the BK=64 kernel's envelope (cin %% 64 != 0 or tensor too large)");
        return OSR_ERR_UNSUPPORTED;
    }
    if (p->in_dtype == OSR_F16) {
        if (p->out_dtype == OSR_F16) return conv_launch<f16_t, f16_t>(a, st);
        if (p->out_dtype == OSR_F32) return conv_launch<f16_t, float>(a, st);
    } else {
        if (p->out_dtype == OSR_BF16) return conv_launch<bf16_t, bf16_t>(a, st);
        if (p->out_dtype == OSR_F32) return conv_launch<bf16_t, float>(a, st);
    }
    osr_set_error("osr_conv2d_fwd: out_dtype must equal in_dtype or be f32");
    return OSR_ERR_UNSUPPORTED;
}

extern "C" int64_t osr_conv2d_fwd_workspace_bytes(const osr_conv_params* p) {
    if (!p || (p->in_dtype != OSR_F16 && p->in_dtype != OSR_BF16) || p->n < 1 || p->ho < 1 || p->wo < 1 || p->cin < 64 || p->cout < 8 || p->cout % 8 != 0) return 0;
    return osr_conv64_split_workspace_bytes(p);
}

extern "C" int32_t osr_conv2d_fwd_describe(const osr_conv_params* p, int32_t has_workspace, char* buf, int32_t buf_bytes) {
    if (!p || !buf || buf_bytes < 1) return OSR_ERR_INVALID_ARG;
    if ((p->in_dtype != OSR_F16 && p->in_dtype != OSR_BF16) || p->cin % 64 != 0 || p->n < 1 || p->ho < 1 || p->wo < 1 || p->cout < 8)
        return snprintf(buf, buf_bytes, "not on the BK=64 kernel");
    return osr_conv64_describe(p, has_workspace, buf, buf_bytes);
}

extern "C" osr_status osr_conv2d_fwd(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual,
                                     void* out, void* stream) {
    return conv2d_fwd_impl(p, in, weight, bias, residual, nullptr, false, out, stream);
}

extern "C" osr_status osr_conv2d_fwd_masked(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual,
                                            const void* mask, void* out, void* stream) {
    return conv2d_fwd_impl(p, in, weight, bias, residual, mask, true, out, stream);
}

// ------------------------------------------------------------------------------------------------------
// exact fp32 GEMM on v_mfma_f32_32x32x2_f32: out[m][n] = sum_k a[m][k]*w[n][k] + bias[n]
// 64x64x16 tile, 4 waves (2x2), LDS k-major so fragment reads are consecutive 4-byte words.
// ------------------------------------------------------------------------------------------------------
#define G32_LD 68

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out, long long ldo, int M, int N, int K,
                                                       int relu) {
    __shared__ float sA[2][16][G32_LD];
    __shared__ float sB[2][16][G32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const bool aok = m0 + lrow < M, bok = n0 + lrow < N;
    const float* ap = A + (long long)(aok ? m0 + lrow : 0) * lda + lk;
    const float* bp = W + (long long)(bok ? n0 + lrow : 0) * K + lk;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra = aok ? *reinterpret_cast<const float4*>(ap) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rb = bok ? *reinterpret_cast<const float4*>(bp) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto stage = [&](int buf) {
        sA[buf][lk + 0][lrow] = ra.x; sA[buf][lk + 1][lrow] = ra.y; sA[buf][lk + 2][lrow] = ra.z; sA[buf][lk + 3][lrow] = ra.w;
        sB[buf][lk + 0][lrow] = rb.x; sB[buf][lk + 1][lrow] = rb.y; sB[buf][lk + 2][lrow] = rb.z; sB[buf][lk + 3][lrow] = rb.w;
    };
    stage(0);
    __syncthreads();
    const int nk = K / 16;
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) {
            ra = aok ? *reinterpret_cast<const float4*>(ap + (ks + 1) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb = bok ? *reinterpret_cast<const float4*>(bp + (ks + 1) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int buf = ks & 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float fa = sA[buf][kk * 2 + (lane >> 5)][wr * 32 + (lane & 31)];
            const float fb = sB[buf][kk * 2 + (lane >> 5)][wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    const int col = n0 + wc * 32 + (lane & 31);
    if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) {
                float v = acc[r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                out[(long long)row * ldo + col] = v;
            }
        }
    }
}

extern "C" osr_status osr_gemm_f32(const float* a, int64_t lda, const float* w, const float* bias, float* out, int64_t ldo, int32_t m,
                                   int32_t n, int32_t k, int32_t relu, void* stream) {
    OSR_REQUIRE(a && w && out, OSR_ERR_INVALID_ARG, "osr_gemm_f32: null pointer");
    OSR_REQUIRE(m >= 0 && n >= 1 && k >= 16 && k % 16 == 0, OSR_ERR_UNSUPPORTED, "osr_gemm_f32: k must be a multiple of 16 (got %d)", k);
    OSR_REQUIRE(lda >= k && lda % 4 == 0 && ldo >= n, OSR_ERR_INVALID_ARG, "osr_gemm_f32: bad leading dimensions");
    OSR_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_gemm_f32: a/w must be 16-byte aligned");
    if (m == 0) return OSR_OK;
    dim3 grid((n + 63) / 64, (m + 63) / 64);
    OSR_REQUIRE(grid.y <= 65535, OSR_ERR_UNSUPPORTED, "osr_gemm_f32: m too large");
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, (long long)lda, w, bias, out, (long long)ldo, m, n, k, relu);
    OSR_CHECK_LAUNCH("osr_gemm_f32");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// out[m][n] = sum_k a[k][m] * b[k][n]: the weight-gradient product dW = dy^T x of an fp32 linear layer, both operands read as
// they lie (row = sample), so no transposed copies. Same 64x64x16 tile and k-major LDS as gemm_f32_kernel (the staging is a
// straight copy here). The sample axis is split over blockIdx.z: each split writes its partial tile to the workspace and
// gemm_f32_tn_reduce adds the splits in index order (a fixed order: results do not depend on scheduling).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_tn_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                          float* __restrict__ out, long long ldo, long long split_stride, int M, int N, int K,
                                                          int kchunk) {
    __shared__ __attribute__((aligned(16))) float sA[2][16][G32_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][16][G32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_lo = blockIdx.z * kchunk, k_hi = min(K, k_lo + kchunk);
    const int lk = tid >> 4, l4 = (tid & 15) * 4;
    const bool avec = (lda & 3) == 0 && m0 + 64 <= M && (((uintptr_t)A) & 15) == 0;
    const bool bvec = (ldb & 3) == 0 && n0 + 64 <= N && (((uintptr_t)B) & 15) == 0;
    auto fetch = [&](const float* P, long long ld, int c0, int C, bool vec, int k) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_hi) {
            const float* p = P + (long long)k * ld + c0 + l4;
            if (vec) v = *reinterpret_cast<const float4*>(p);
            else {
                if (c0 + l4 + 0 < C) v.x = p[0];
                if (c0 + l4 + 1 < C) v.y = p[1];
                if (c0 + l4 + 2 < C) v.z = p[2];
                if (c0 + l4 + 3 < C) v.w = p[3];
            }
        }
        return v;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra = fetch(A, lda, m0, M, avec, k_lo + lk), rb = fetch(B, ldb, n0, N, bvec, k_lo + lk);
    *reinterpret_cast<float4*>(&sA[0][lk][l4]) = ra;
    *reinterpret_cast<float4*>(&sB[0][lk][l4]) = rb;
    __syncthreads();
    const int nk = (k_hi - k_lo + 15) / 16;
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) {
            ra = fetch(A, lda, m0, M, avec, k_lo + (ks + 1) * 16 + lk);
            rb = fetch(B, ldb, n0, N, bvec, k_lo + (ks + 1) * 16 + lk);
        }
        const int buf = ks & 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float fa = sA[buf][kk * 2 + (lane >> 5)][wr * 32 + (lane & 31)];
            const float fb = sB[buf][kk * 2 + (lane >> 5)][wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
        if (more) {
            *reinterpret_cast<float4*>(&sA[buf ^ 1][lk][l4]) = ra;
            *reinterpret_cast<float4*>(&sB[buf ^ 1][lk][l4]) = rb;
        }
        __syncthreads();
    }
    out += (long long)blockIdx.z * split_stride;
    const int col = n0 + wc * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) out[(long long)row * ldo + col] = acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_tn_reduce(const float* __restrict__ ws, long long split_stride, int splits, float* __restrict__ out,
                                                          long long ldo, int M, int N) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * N) return;
    float s = ws[i];
    for (int k = 1; k < splits; ++k) s += ws[k * split_stride + i];
    out[(i / N) * ldo + i % N] = s;
}

static int gemm_tn_splits(int m, int n, int k) {
    const long long tiles = (long long)((m + 63) / 64) * ((n + 63) / 64);
    long long s = (768 + tiles - 1) / tiles;                 // ~3 workgroups per CU
    s = std::min<long long>(s, std::max(1, k / 128));        // at least 128 samples per split
    return (int)std::max<long long>(1, std::min<long long>(s, 64));
}

extern "C" int64_t osr_gemm_f32_tn_workspace_bytes(int32_t m, int32_t n, int32_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int s = gemm_tn_splits(m, n, k);
    return s > 1 ? (int64_t)s * m * n * 4 : 0;
}

extern "C" osr_status osr_gemm_f32_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int32_t m, int32_t n,
                                      int32_t k, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(a && b && out, OSR_ERR_INVALID_ARG, "osr_gemm_f32_tn: null pointer");
    OSR_REQUIRE(m >= 1 && n >= 1 && k >= 0, OSR_ERR_INVALID_ARG, "osr_gemm_f32_tn: bad shape");
    OSR_REQUIRE(lda >= m && ldb >= n && ldo >= n, OSR_ERR_INVALID_ARG, "osr_gemm_f32_tn: bad leading dimensions");
    int splits = gemm_tn_splits(m, n, std::max(k, 1));
    const int64_t per = (int64_t)m * n * 4;
    if (!workspace || workspace_bytes < 2 * per) splits = 1;
    else splits = (int)std::min<int64_t>(splits, workspace_bytes / per);
    int kchunk = ((k + splits - 1) / splits + 15) / 16 * 16;
    if (kchunk < 16) kchunk = 16;
    splits = std::max(1, (k + kchunk - 1) / kchunk);
    dim3 grid((n + 63) / 64, (m + 63) / 64, splits);
    OSR_REQUIRE(grid.y <= 65535, OSR_ERR_UNSUPPORTED, "osr_gemm_f32_tn: m too large");
    if (splits == 1) {
        hipLaunchKernelGGL(gemm_f32_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, (long long)lda, b, (long long)ldb, out, (long long)ldo, 0LL, m, n,
                           k, kchunk);
        OSR_CHECK_LAUNCH("osr_gemm_f32_tn");
        return OSR_OK;
    }
    hipLaunchKernelGGL(gemm_f32_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, (long long)lda, b, (long long)ldb, (float*)workspace, (long long)n,
                       (long long)m * n, m, n, k, kchunk);
    OSR_CHECK_LAUNCH("osr_gemm_f32_tn");
    const long long tot = (long long)m * n;
    hipLaunchKernelGGL(gemm_f32_tn_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, tot, splits, out,
                       (long long)ldo, m, n);
    OSR_CHECK_LAUNCH("osr_gemm_f32_tn_reduce");
    return OSR_OK;
}
