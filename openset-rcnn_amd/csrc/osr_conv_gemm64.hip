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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

template <class T> struct Frag64;
template <> struct Frag64<f16_t> {
    typedef f16x8 type;
    static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Frag64<bf16_t> {
    typedef bf16x8 type;
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

struct Conv64Args {
    osr_conv_params p;
    const void* in;
    const void* w;
    const float* bias;
    const void* res;
    void* out;
    long long M;
    int K;                 // GEMM K of the weight rows (multiple of 64)
    int tiles_m, tiles_n;
    unsigned in_bytes, w_bytes;  // buffer sizes for the bounds check (< 2 GiB)
    int stem;              // 1: cin == 32 view, two taps per K slice
};

#define OOB_OFF 0x80000000u

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

template <class TI, class TO, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm64_kernel(Conv64Args a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int A_PIECES = BM / 8 / 4, B_PIECES = BN / 8 / 4;  // 1-KiB LDS-DMA pieces per wave per K step
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int EPI_LD = TN * 32 + 4;
    constexpr int EPI_BYTES = 4 * 32 * EPI_LD * 4;
    constexpr int LDS_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    typedef typename Frag64<TI>::type frag_t;
    const osr_conv_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;

    // XCD-aware bijective remap of the linear block id (guide T1)
    const int nwg = a.tiles_m * a.tiles_n;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = t % a.tiles_n, tile_m = t / a.tiles_n;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int howo = p.ho * p.wo;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

    // ---- per-lane gather descriptors: this lane serves row (piece*8 + lane/8), LDS slot lane%8 ----
    const int lrow = lane >> 3, slot = lane & 7;
    unsigned a_base[A_PIECES];  // byte offset of the pixel row's (n, 0, 0) origin + chunk, or OOB
    int a_ih0[A_PIECES], a_iw0[A_PIECES];
    unsigned a_chunk[A_PIECES];  // logical chunk (0..7) this lane fetches
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int row = (wid * A_PIECES + j) * 8 + lrow;
        const long long m = m0 + row;
        const bool ok = m < a.M;
        const long long mm = ok ? m : 0;
        const int nimg = (int)(mm / howo), rem = (int)(mm - (long long)nimg * howo);
        const int oh = rem / p.wo, ow = rem - oh * p.wo;
        a_ih0[j] = oh * p.stride_h - p.pad_h;
        a_iw0[j] = ow * p.stride_w - p.pad_w;
        a_chunk[j] = (unsigned)(slot ^ ((row >> 1) & 7));
        a_base[j] = ok ? (unsigned)((long long)nimg * p.in_stride_n * 2) : OOB_OFF;
    }
    unsigned b_off[B_PIECES];
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) {
        const int row = (wid * B_PIECES + j) * 8 + lrow;
        const unsigned chunk = (unsigned)(slot ^ ((row >> 1) & 7));
        const long long o = ((long long)(n0 + row) * a.K + chunk * 8) * 2;
        b_off[j] = (n0 + row < p.cout && o < (long long)OOB_OFF) ? (unsigned)o : OOB_OFF;
    }

    int kh = 0, kw = 0, c0 = 0;  // tap / channel origin of the current K slice (non-stem)
    int kbyte = 0;               // byte offset of the K slice inside a weight row

#define C64_ISSUE(stage)                                                                                                         \
    {                                                                                                                            \
        unsigned char* sa_ = lds + (stage) * STAGE;                                                                              \
        unsigned char* sb_ = sa_ + BM * 128;                                                                                     \
        _Pragma("unroll") for (int j = 0; j < A_PIECES; ++j) {                                                                   \
            int ih, iw, cc;                                                                                                      \
            if (a.stem) { /* two 32-wide taps per slice: chunks 0-3 -> tap kh, 4-7 -> tap kh+1 */                                \
                ih = a_ih0[j] + kh + (int)(a_chunk[j] >> 2); iw = a_iw0[j]; cc = (int)(a_chunk[j] & 3) * 8;                       \
            } else { ih = a_ih0[j] + kh; iw = a_iw0[j] + kw; cc = c0 + (int)a_chunk[j] * 8; }                                     \
            bool ok = a_base[j] != OOB_OFF;                                                                                      \
            if (p.pad_mode == 0) ok = ok && (unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi;                      \
            const unsigned off = ok ? a_base[j] + (unsigned)(((long long)ih * p.in_stride_h + (long long)iw * p.in_stride_w + cc) * 2) : OOB_OFF; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t*)(sa_ + (wid * A_PIECES + j) * 1024), 16, off, 0, 0, 0); \
        }                                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < B_PIECES; ++j) {                                                                   \
            const unsigned off = b_off[j] == OOB_OFF ? OOB_OFF : b_off[j] + (unsigned)kbyte;                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(sb_ + (wid * B_PIECES + j) * 1024), 16, off, 0, 0, 0);  \
        }                                                                                                                        \
    }
#define C64_ADVANCE()                                                            \
    {                                                                            \
        kbyte += 128;                                                            \
        if (a.stem) kh += 2;                                                     \
        else { c0 += 64; if (c0 >= p.cin) { c0 = 0; if (++kw >= p.kw) { kw = 0; ++kh; } } } \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((lane & 31) >> 1) & 7;
    const int nk = a.K / 64;
    C64_ISSUE(0);
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile ks landed for every wave; every wave is done reading the other stage
        if (ks + 1 < nk) { C64_ADVANCE(); C64_ISSUE((ks + 1) & 1); }
        const unsigned char* sa = lds + (ks & 1) * STAGE;
        const unsigned char* sb = sa + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int sl = ((kk * 2 + (lane >> 5)) ^ swz) * 16;
            frag_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const frag_t*>(sa + ((wr * TM + i) * 32 + (lane & 31)) * 128 + sl);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const frag_t*>(sb + ((wc * TN + j) * 32 + (lane & 31)) * 128 + sl);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Frag64<TI>::mfma(fa[i], fb[j], acc[i][j]);
        }
    }
    __syncthreads();  // all waves done with the staging buffers before the epilogue reuses them

    // ---- epilogue (same scheme as the BK=32 kernel): wave-private fp32 slab -> 8 channels per lane ----
    float* slab = reinterpret_cast<float*>(lds) + wid * 32 * EPI_LD;
    TO* __restrict__ out = reinterpret_cast<TO*>(a.out);
    const TI* __restrict__ res = reinterpret_cast<const TI*>(a.res);
    constexpr int LPR = TN * 4, RPP = 64 / LPR;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                slab[row * EPI_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
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
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                store8_64<TO>(out + (long long)nimg * p.out_stride_n + (long long)oh * p.out_stride_h + (long long)ow * p.out_stride_w + co, v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <class TI, class TO>
static osr_status conv64_launch(Conv64Args& a, hipStream_t st) {
    if (a.p.cout <= 64) {
        a.tiles_m = (int)((a.M + 127) / 128);
        a.tiles_n = (a.p.cout + 63) / 64;
        hipLaunchKernelGGL((conv_igemm64_kernel<TI, TO, 128, 64, 4, 1>), dim3((unsigned)a.tiles_m * a.tiles_n), dim3(256), 0, st, a);
    } else {
        a.tiles_m = (int)((a.M + 127) / 128);
        a.tiles_n = (a.p.cout + 127) / 128;
        hipLaunchKernelGGL((conv_igemm64_kernel<TI, TO, 128, 128, 2, 2>), dim3((unsigned)a.tiles_m * a.tiles_n), dim3(256), 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { osr_set_error("osr_conv2d_fwd(bk64): launch failed: %s", hipGetErrorString(e)); return OSR_ERR_LAUNCH; }
    return OSR_OK;
}

// Returns 1 when this fast path can take the problem (all arguments already validated by osr_conv2d_fwd).
int osr_conv64_eligible(const osr_conv_params* p, long long in_bytes, long long w_bytes) {
    const bool stem = p->pad_mode == 1 && p->cin == 32 && p->kw == 1 && (p->kh % 2) == 0;
    if (!stem && p->cin % 64 != 0) return 0;
    if (in_bytes <= 0 || w_bytes <= 0 || in_bytes >= (1ll << 31) - 4096 || w_bytes >= (1ll << 31) - 4096) return 0;
    return 1;
}

osr_status osr_conv64_run(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual, void* out,
                          long long in_bytes, long long w_bytes, hipStream_t st) {
    Conv64Args a;
    a.p = *p; a.in = in; a.w = weight; a.bias = bias; a.res = residual; a.out = out;
    a.M = (long long)p->n * p->ho * p->wo;
    a.K = p->kh * p->kw * p->cin;
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
    a.stem = (p->pad_mode == 1 && p->cin == 32) ? 1 : 0;
    a.tiles_m = a.tiles_n = 0;
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
