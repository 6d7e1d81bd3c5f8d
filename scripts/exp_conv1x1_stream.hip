// EXPERIMENT, not part of the product build (round 3; DESIGN.md section 3, "the 128 x 128 bucket"): a persistent, continuously
// double-buffered 1x1 convolution for the res3-res5 conv1 / conv3 / shortcut layers. It was hooked into osr_conv2d_fwd, passed its
// parity test (bit-identical to conv_igemm64_kernel on nine layer shapes x two dtypes, deterministic between other launches) and
// was measured against the tile-model kernel on one box, two interleaved bench runs each: conv family 10.51 / 10.57 ms with it,
// 10.40 / 10.35 ms without; per layer: conv3 + residual 2-4 % faster (res3 136 -> 133 us, res4 90 -> 86), res3 conv1 8 % faster,
// the stride-2 shortcuts 16-25 % slower, res5 conv1 (K = 2048) 34 % slower, res4 conv1 6 % slower. The entry latency the stamps
// showed (6 of 15 us per workgroup) is queueing in a memory system that is already saturated at 3.5-4.6 TB/s by these access
// patterns, not an idle pipe: requesting a tile's first slice a compute phase earlier moves the wait, it does not remove it.
// Kept here for the record; to rebuild the experiment add it to build.py's SOURCES and route eligible layers in osr_conv2d_fwd.
//
// 1x1 convolutions of the bottleneck stages (res3-res5 conv1 / conv3 / shortcut) as a PERSISTENT, continuously double-buffered
// GEMM stream (round 3). osr_conv2d_fwd routes a layer here when it is a 1x1 convolution without padding whose cin is a multiple
// of 64 and cout a multiple of 128, with no residual or the plain one (res_mode 1), in_dtype == out_dtype (f16 / bf16).
//
// Why a second kernel for these layers: in-kernel stamps of conv_igemm64_kernel (scripts/exp_conv_stamps.py) show that a
// workgroup of res4.conv3 lives 15 us of which 6 us pass between its entry and the landing of its first K slice (res3.conv3: 5.6 of
// 11.4 us) -- the tile's first slice and its residual are requested only when the workgroup starts, the K loop is 2-16 slices short,
// and the single staging buffer is refilled only after the MFMAs that read it. With three workgroups per CU the chip then keeps
// ~16 MB in flight and these HBM-bound layers run at 3.2-4.5 TB/s. Here a workgroup never starts cold twice:
//  * persistent workgroups (two per CU) walk the tiles  L, L + G, L + 2G, ...  (XCD-aware order: the N tiles of an M tile are
//    neighbours on one XCD, so the activation rows reach that L2 once);
//  * ONE ring of two 32-KB stages runs across K steps AND tile boundaries: while stage s is multiplied, stage s + 1 -- the next K
//    slice, or slice 0 of the workgroup's NEXT tile -- is on its way (LDS-DMA), and the next tile's residual and bias are requested
//    in front of it; a tile's first MFMA waits for nothing that was not requested a whole compute phase earlier;
//  * swapped MFMA operand roles (weights = A, pixels = B; weight rows permuted per 64-row block as in osr_bottleneck.hip): a lane
//    ends up with eight consecutive output channels of one pixel, so the epilogue is registers -> bias -> residual (16-byte load at
//    the address of the store) -> ReLU -> 16-byte store, with NO LDS slab: the staging ring is never interrupted by an epilogue.
// K order (k ascending, two 32-wide MFMA steps per 64-wide slice) and rounding are those of conv_igemm64_kernel: results are
// bit-identical (tests/test_conv1x1.py).
#include "osr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f16_t c1f16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t c1bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned c1u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void c1_lds_void_t;

template <class T> struct C1Frag;
template <> struct C1Frag<f16_t> {
    typedef c1f16x8 type;
    static __device__ __forceinline__ f32x4 mfma(c1f16x8 a, c1f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct C1Frag<bf16_t> {
    typedef c1bf16x8 type;
    static __device__ __forceinline__ f32x4 mfma(c1bf16x8 a, c1bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

struct C1Div {
    unsigned mp, sh1, sh2, d;
};
static C1Div c1div_make(unsigned d) {
    C1Div f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mp = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    f.d = d;
    return f;
}
__device__ __forceinline__ unsigned c1div(unsigned n, const C1Div& f) {
    const unsigned t = __umulhi(f.mp, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct C1Args {
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    void* out;
    long long M;
    int K, cout, relu;
    int tiles_m, tiles_n;
    int stride_h, stride_w;                 // of the input sampling (1 or 2)
    long long in_stride_n, in_stride_h, in_stride_w;  // elements
    C1Div div_howo, div_wo;                 // output row -> (image, oh, ow)
    int dense_in;                           // 1: input row m starts at element m * K (stride 1, contiguous NHWC)
    unsigned in_bytes, w_bytes, out_bytes;
};

#define C1_BM 128
#define C1_BN 128
#define C1_STAGE 32768  // A slice 128 rows x 128 B + B slice 128 rows x 128 B
#define C1_OOB 0x80000000u

// LDS row rho of a 64-row weight block holds weight row n(T, i) = 32 (T >> 1) + 8 (i >> 2) + 4 (T & 1) + (i & 3), T = rho >> 4, i = rho & 15:
// the lane that holds rows 4g..4g+3 of the output tiles 2p and 2p+1 then holds channels 32p + 8g .. + 7 (osr_bottleneck.hip).
__device__ __forceinline__ int c1_nmap(int rho) {
    const int T = rho >> 4, i = rho & 15;
    return 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);
}

template <class TI> __device__ __forceinline__ typename C1Frag<TI>::type c1_as_frag(c1u32x4 v) {
    union { c1u32x4 u; typename C1Frag<TI>::type f; } c;
    c.u = v;
    return c.f;
}
__device__ __forceinline__ c1f16x8 c1_relu_pack(const float v[8], int relu, f16_t) {
    c1f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16_t)v[e];
    const c1f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    return relu ? __builtin_elementwise_max(o, zero) : o;  // (rounding is monotonic and maps 0 to 0: relu commutes with it)
}
__device__ __forceinline__ c1bf16x8 c1_relu_pack(const float v[8], int relu, bf16_t) {
    c1bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(relu ? fmaxf(v[e], 0.f) : v[e]);
    return o;
}

// tile of linear id L: XCD-aware bijective remap (ids that share L & 7 run on one XCD under round-robin placement and walk a
// contiguous run of tiles, the N tiles of one M tile adjacent)
__device__ __forceinline__ void c1_tile(int L, int ntiles, int tiles_n, int& m0, int& n0) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = L & 7, idx = L >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tn = t % tiles_n, tm = t / tiles_n;
    m0 = tm * C1_BM;
    n0 = tn * C1_BN;
}

template <class TI, int RES>
__global__ __launch_bounds__(256, 2) void conv1x1s_kernel(C1Args a) {
    typedef typename C1Frag<TI>::type frag_t;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // 2 stages of 32 KB
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;  // this wave: pixels 64 wr .. + 63, channels 64 wc .. + 63 of the tile
    const int ntiles = a.tiles_m * a.tiles_n, nk = a.K >> 6;
    const int G = gridDim.x;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? a.res : a.out), 0, a.out_bytes, 0x00020000);

#define C1_SW(row, chunk) ((row) * 128 + ((((chunk)) ^ (((row) >> 1) & 7)) << 4))

    // ---- per-tile staging descriptors of this lane: 4 activation pieces + 4 weight pieces of 1 KiB per stage (piece = 8 rows x 128 B;
    //      lane = LDS row piece*8 + lane/8, slot lane%8, which receives logical chunk slot ^ ((row >> 1) & 7)) ----
    unsigned a_off[4], b_off[4];  // byte offsets of K slice 0 (OOB: zero fill)
    int lane = lane0, l15 = lane0 & 15, g = lane0 >> 4;
#define C1_SETUP(m0_, n0_)                                                                                                          \
    {                                                                                                                               \
        const int lrow_ = lane >> 3, slot_ = lane & 7;                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                          \
            const int row_ = (wid * 4 + j_) * 8 + lrow_;                                                                            \
            const unsigned chunk_ = (unsigned)(slot_ ^ ((row_ >> 1) & 7));                                                          \
            const long long m_ = (long long)(m0_) + row_;                                                                           \
            long long e_;                                                                                                           \
            if (a.dense_in) e_ = m_ * a.K;                                                                                          \
            else {                                                                                                                  \
                const unsigned mu_ = (unsigned)(m_ < a.M ? m_ : 0);                                                                 \
                const unsigned ni_ = c1div(mu_, a.div_howo), rem_ = mu_ - ni_ * a.div_howo.d;                                        \
                const unsigned oh_ = c1div(rem_, a.div_wo), ow_ = rem_ - oh_ * a.div_wo.d;                                           \
                e_ = (long long)ni_ * a.in_stride_n + (long long)oh_ * a.stride_h * a.in_stride_h + (long long)ow_ * a.stride_w * a.in_stride_w; \
            }                                                                                                                       \
            a_off[j_] = m_ < a.M ? (unsigned)((e_ + chunk_ * 8) * 2) : C1_OOB;                                                      \
            const int rho_ = row_ & 63, blk_ = row_ >> 6; /* weight rows: permuted inside each 64-row block */                       \
            b_off[j_] = (unsigned)((((long long)(n0_) + blk_ * 64 + c1_nmap(rho_)) * a.K + chunk_ * 8) * 2);                         \
        }                                                                                                                           \
    }
#define C1_ISSUE(stage_, ks_)                                                                                                       \
    {                                                                                                                               \
        unsigned char* sa_ = lds + (stage_) * C1_STAGE;                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                          \
            const unsigned ao_ = a_off[j_] == C1_OOB ? C1_OOB : a_off[j_] + (unsigned)(ks_) * 128u;                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (c1_lds_void_t*)(sa_ + (wid * 4 + j_) * 1024), 16, ao_, 0, 0, 0);       \
        }                                                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (c1_lds_void_t*)(sa_ + 16384 + (wid * 4 + j_) * 1024), 16, b_off[j_] + (unsigned)(ks_) * 128u, 0, 0, 0); \
    }
    // epilogue operands of a tile, requested ahead: residual chunks [pixel tile pt][channel pair p] (the address of the store) and the
    // bias of this lane's 2 x 8 channels
#define C1_FETCH_BIAS(n0_)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                                          \
            const int c_ = (n0_) + wc * 64 + 32 * p_ + 8 * g;                                                                       \
            const float4 b0_ = *reinterpret_cast<const float4*>(a.bias + c_), b1_ = *reinterpret_cast<const float4*>(a.bias + c_ + 4); \
            bias_c[p_][0] = b0_.x; bias_c[p_][1] = b0_.y; bias_c[p_][2] = b0_.z; bias_c[p_][3] = b0_.w;                             \
            bias_c[p_][4] = b1_.x; bias_c[p_][5] = b1_.y; bias_c[p_][6] = b1_.z; bias_c[p_][7] = b1_.w;                             \
        }                                                                                                                           \
    }
#define C1_FETCH_EPI(O_, R_, m0_, n0_)                                                                                              \
    {                                                                                                                               \
        _Pragma("unroll") for (int pt_ = 0; pt_ < 4; ++pt_) {                                                                       \
            const long long m_ = (long long)(m0_) + wr * 64 + pt_ * 16 + l15;                                                       \
            O_[pt_] = m_ < a.M ? (unsigned)((m_ * a.cout + (n0_) + wc * 64 + 8 * g) * 2) : C1_OOB;                                   \
            if constexpr (RES) {                                                                                                    \
                _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_)                                                                    \
                    R_[pt_][p_] = c1_as_frag<TI>(__builtin_amdgcn_raw_buffer_load_b128(rs_res, O_[pt_] == C1_OOB ? C1_OOB : O_[pt_] + 64u * p_, 0, 0)); \
            }                                                                                                                       \
        }                                                                                                                           \
    }

    // (two NAMED sets, current and next, copied at the tile boundary: an array indexed by a run-time buffer number lives in scratch)
    float bias_c[2][8];  // (this tile's bias: an L2 hit, requested in the tile's last K step)
    unsigned ooff_c[4], ooff_n[4];
    frag_t rres_c[RES ? 4 : 1][RES ? 2 : 1], rres_n[RES ? 4 : 1][RES ? 2 : 1];

    int L = blockIdx.x, m0, n0;
    if (L >= ntiles) return;
    c1_tile(L, ntiles, a.tiles_n, m0, n0);
    C1_FETCH_EPI(ooff_c, rres_c, m0, n0);
    asm volatile("" ::: "memory");  // (issue order is what the counted wait below relies on: epilogue operands, then the stage's pieces)
    C1_SETUP(m0, n0);
    C1_ISSUE(0, 0);
    int stage = 0;
    bool first = true;

    for (;;) {
        const int Ln = L + G;
        const bool has_next = Ln < ntiles;
        int m0n = 0, n0n = 0;
        if (has_next) c1_tile(Ln, ntiles, a.tiles_n, m0n, n0n);
        // (the lane id is laundered once per tile: the address arithmetic of the body then depends on a value defined inside the loop
        // and cannot be hoisted out of it into registers that stay live across iterations)
        {
            int lv = lane0;
            asm volatile("" : "+v"(lv));
            lane = lv & 63; l15 = lane & 15; g = lane >> 4;
        }
        f32x4 acc[4][4];  // [pixel tile][channel tile]
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[pt][T] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ks = 0; ks < nk; ++ks) {
            // RAW: this stage's pieces have landed for this wave when at most the operations issued after them are outstanding: the 8
            // stores of the previous tile's epilogue in front of a tile's first stage, nothing otherwise; and for every wave behind
            // the barrier. WAR: the stage refilled below was read in the previous step; lgkmcnt(0) retires this wave's fragment reads
            // of it before any wave's LDS-DMA may land there (the rule the fused res2 block learned the hard way).
            if (ks == 0 && !first) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const unsigned char* sa = lds + stage * C1_STAGE;
            const unsigned char* sb = sa + 16384;
            // the next stage: the next K slice of this tile, or -- with the next tile's epilogue operands in front of it -- slice 0 of
            // the workgroup's next tile
            if (ks + 1 < nk) {
                C1_ISSUE(stage ^ 1, ks + 1);
            } else {
                C1_FETCH_BIAS(n0);
                if (has_next) {
                    C1_FETCH_EPI(ooff_n, rres_n, m0n, n0n);
                    asm volatile("" ::: "memory");
                    C1_SETUP(m0n, n0n);
                    C1_ISSUE(stage ^ 1, 0);
                }
            }
#pragma unroll
            for (int k32 = 0; k32 < 2; ++k32) {
                frag_t wf[4], pf[4];
#pragma unroll
                for (int T = 0; T < 4; ++T) wf[T] = *reinterpret_cast<const frag_t*>(sb + C1_SW(wc * 64 + T * 16 + l15, k32 * 4 + g));
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) pf[pt] = *reinterpret_cast<const frag_t*>(sa + C1_SW(wr * 64 + pt * 16 + l15, k32 * 4 + g));
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int T = 0; T < 4; ++T) acc[pt][T] = C1Frag<TI>::mfma(wf[T], pf[pt], acc[pt][T]);
            }
            stage ^= 1;
        }
        first = false;
        // ---- epilogue from registers: lane (l15, g) holds channels 32p + 8g .. + 7 of pixel pt*16 + l15 in acc[pt][2p], acc[pt][2p+1] ----
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = (e < 4 ? acc[pt][2 * p][e] : acc[pt][2 * p + 1][e - 4]) + bias_c[p][e];
                    if constexpr (RES) v[e] += (float)rres_c[pt][p][e];
                }
                union { frag_t f; c1u32x4 u; } cv;
                cv.f = c1_relu_pack(v, a.relu, TI());
                __builtin_amdgcn_raw_buffer_store_b128(cv.u, rs_out, ooff_c[pt] == C1_OOB ? C1_OOB : ooff_c[pt] + 64u * p, 0, 0);
            }
        if (!has_next) break;
        L = Ln; m0 = m0n; n0 = n0n;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            ooff_c[pt] = ooff_n[pt];
            if constexpr (RES) {
#pragma unroll
                for (int p = 0; p < 2; ++p) rres_c[pt][p] = rres_n[pt][p];
            }
        }
    }
}

// Returns 1 when this kernel takes the layer (all arguments already validated by osr_conv2d_fwd).
int osr_conv1x1s_eligible(const osr_conv_params* p, const void* mask) {
    if (mask || p->row_seg_counts) return 0;
    if (p->kh != 1 || p->kw != 1 || p->pad_h != 0 || p->pad_w != 0 || p->pad_mode != 0) return 0;
    if (p->cin % 64 != 0 || p->cin < 128 || p->cout % 128 != 0) return 0;
    if (p->res_mode != 0 && p->res_mode != 1) return 0;
    if (!(p->in_dtype == OSR_F16 || p->in_dtype == OSR_BF16) || p->out_dtype != p->in_dtype) return 0;
    if (p->stride_h < 1 || p->stride_h > 2 || p->stride_w != p->stride_h) return 0;
    const long long M = (long long)p->n * p->ho * p->wo;
    // dense output (and residual) rows: the epilogue addresses row m at m * cout
    if (p->out_stride_w != p->cout || p->out_stride_h != (long long)p->wo * p->cout || p->out_stride_n != (long long)p->ho * p->wo * p->cout) return 0;
    if (p->res_mode == 1 && (p->res_stride_w != p->cout || p->res_stride_h != (long long)p->wo * p->cout || p->res_stride_n != (long long)p->ho * p->wo * p->cout)) return 0;
    if (p->in_stride_w != p->cin) return 0;  // (a pixel's channels are contiguous; rows / images may be strided)
    if (M < 4096 || M >= (1ll << 31) - 1024) return 0;  // small problems keep the tile-model kernel (FC layers of a few hundred rows, tests)
    if (M * p->cout * 2 >= (1ll << 31) - 4096 || (long long)p->n * p->in_stride_n * 2 >= (1ll << 31) - 4096) return 0;
    // HBM-bound shapes only: few K slices per tile. Deep-K layers (FC1, FC2) are MFMA-bound and keep the 256 x 256 tiles.
    if (p->cin > 2048) return 0;
    return 1;
}

template <class TI>
static osr_status conv1x1s_launch(C1Args& a, int res, hipStream_t st) {
    static std::atomic<int> n_cu{0};
    int cus = n_cu.load(std::memory_order_relaxed);
    if (cus <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu.store(cus, std::memory_order_relaxed);
    }
    const long long ntiles = (long long)a.tiles_m * a.tiles_n;
    const unsigned grid = (unsigned)(ntiles < 2ll * cus ? ntiles : 2ll * cus);
    const size_t ldsb = 2 * C1_STAGE;
    if (res) {
        static osr_dev_mask m{0};
        osr_once_per_device(m, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1s_kernel<TI, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C1_STAGE); });
        hipLaunchKernelGGL((conv1x1s_kernel<TI, 1>), dim3(grid), dim3(256), ldsb, st, a);
    } else {
        static osr_dev_mask m{0};
        osr_once_per_device(m, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1s_kernel<TI, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C1_STAGE); });
        hipLaunchKernelGGL((conv1x1s_kernel<TI, 0>), dim3(grid), dim3(256), ldsb, st, a);
    }
    OSR_CHECK_LAUNCH("osr_conv2d_fwd(1x1 stream)");
    return OSR_OK;
}

osr_status osr_conv1x1s_run(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual, void* out, hipStream_t st) {
    C1Args a;
    a.in = in; a.w = weight; a.bias = bias; a.res = residual; a.out = out;
    a.M = (long long)p->n * p->ho * p->wo;
    a.K = p->cin; a.cout = p->cout; a.relu = p->relu;
    a.tiles_m = (int)((a.M + C1_BM - 1) / C1_BM); a.tiles_n = p->cout / C1_BN;
    a.stride_h = p->stride_h; a.stride_w = p->stride_w;
    a.in_stride_n = p->in_stride_n; a.in_stride_h = p->in_stride_h; a.in_stride_w = p->in_stride_w;
    a.div_howo = c1div_make((unsigned)(p->ho * p->wo));
    a.div_wo = c1div_make((unsigned)p->wo);
    a.dense_in = (p->stride_h == 1 && p->in_stride_h == (long long)p->wi * p->cin && p->in_stride_n == (long long)p->hi * p->wi * p->cin) ? 1 : 0;
    a.in_bytes = (unsigned)((long long)p->n * p->in_stride_n * 2);
    a.w_bytes = (unsigned)((long long)p->cout * p->cin * 2);
    a.out_bytes = (unsigned)(a.M * p->cout * 2);
    return p->in_dtype == OSR_F16 ? conv1x1s_launch<f16_t>(a, p->res_mode == 1, st) : conv1x1s_launch<bf16_t>(a, p->res_mode == 1, st);
}
