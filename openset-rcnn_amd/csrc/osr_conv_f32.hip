// fp32 implicit-GEMM convolution / FC for the parity mode of the engine (include/osr.h: osr_conv2d_fwd with in_dtype OSR_F32).
//
// The reference computes the whole path in fp32 (it never leaves torch's default dtype: /root/reference/train.py:189 builds the
// model, no autocast anywhere), and BASELINE.json's north_star asks for "fp32 box/score/embedding within 1e-4" of it. The fast
// path stores activations in fp16/bf16 (osr_conv_gemm64.hip); this file is the same layer set with fp32 storage and fp32 products:
// activations NHWC fp32, weights [cout][kh][kw][cin] fp32, v_mfma_f32_32x32x2_f32 (an exact fp32 FMA chain in k order, 1/16 of
// the fp16 MFMA rate -- a verification mode, not the benchmark path). Same epilogue semantics as the fast kernels: bias, residual
// add (res_mode 1), FPN nearest-2x upsample-add (res_mode 2), ReLU mask (res_mode 3), ReLU.
//
// 64 x 64 x 16 tile per 256-thread workgroup (2 x 2 waves of 32 x 32), LDS k-major double buffer, register-staged prefetch of the
// next K slice, one barrier per slice. A 16-float K slice is 64 contiguous bytes of one tap of one input pixel (cin % 16 == 0;
// the 7x7 stem runs as the same (kh = 8, kw = 1, cin = 32) view of the pre-padded NHWC4 image the fp16 path uses).
#include "osr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CF_LD 68

struct ConvF32Args {
    osr_conv_params p;
    const float* in;
    const float* w;
    const float* bias;
    const float* res;
    float* out;
    long long M;
    int K, tiles_n;
};

__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF32Args a) {
    __shared__ float sA[2][16][CF_LD];
    __shared__ float sB[2][16][CF_LD];
    const osr_conv_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const long long tile = blockIdx.x;
    const long long m0 = (tile / a.tiles_n) * 64;
    const int n0 = (int)(tile % a.tiles_n) * 64;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const long long howo = (long long)p.ho * p.wo;

    // this thread stages 4 consecutive K elements of activation row m0 + lrow and of weight row n0 + lrow
    const long long m = m0 + lrow;
    const bool aok = m < a.M;
    const long long mm = aok ? m : 0;
    const int nimg = (int)(mm / howo);
    const int rem = (int)(mm - (long long)nimg * howo);
    const int oh = rem / p.wo, ow = rem - oh * p.wo;
    const int ih0 = oh * p.stride_h - p.pad_h, iw0 = ow * p.stride_w - p.pad_w;
    const float* abase = a.in + (long long)nimg * p.in_stride_n + lk;
    const bool bok = n0 + lrow < p.cout;
    const float* bp = a.w + (long long)(bok ? n0 + lrow : 0) * a.K + lk;

    int kh = 0, kw = 0, c0 = 0;  // tap / channel origin of the K slice being loaded
    auto load_a = [&]() -> float4 {
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = aok && (p.pad_mode == 1 || ((unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi));
        return ok ? *reinterpret_cast<const float4*>(abase + (long long)ih * p.in_stride_h + (long long)iw * p.in_stride_w + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto advance = [&]() {
        c0 += 16;
        if (c0 >= p.cin) { c0 = 0; if (++kw >= p.kw) { kw = 0; ++kh; } }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra = load_a();
    float4 rb = bok ? *reinterpret_cast<const float4*>(bp) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto stage = [&](int buf) {
        sA[buf][lk + 0][lrow] = ra.x; sA[buf][lk + 1][lrow] = ra.y; sA[buf][lk + 2][lrow] = ra.z; sA[buf][lk + 3][lrow] = ra.w;
        sB[buf][lk + 0][lrow] = rb.x; sB[buf][lk + 1][lrow] = rb.y; sB[buf][lk + 2][lrow] = rb.z; sB[buf][lk + 3][lrow] = rb.w;
    };
    stage(0);
    __syncthreads();
    const int nk = a.K / 16;
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) {
            advance();
            ra = load_a();
            rb = bok ? *reinterpret_cast<const float4*>(bp + (long long)(ks + 1) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    if (col >= p.cout) return;
    const float bv = a.bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        const int ni = (int)(row / howo);
        const int rm = (int)(row - (long long)ni * howo);
        const int yo = rm / p.wo, xo = rm - yo * p.wo;
        float v = acc[r] + bv;
        if (p.res_mode != 0) {
            const int rh = p.res_mode == 2 ? (yo >> 1) : yo, rw = p.res_mode == 2 ? (xo >> 1) : xo;
            const float rv = a.res[(long long)ni * p.res_stride_n + (long long)rh * p.res_stride_h + (long long)rw * p.res_stride_w + col];
            v = p.res_mode == 3 ? (rv > 0.f ? v : 0.f) : v + rv;
        }
        if (p.relu) v = fmaxf(v, 0.f);
        a.out[(long long)ni * p.out_stride_n + (long long)yo * p.out_stride_h + (long long)xo * p.out_stride_w + col] = v;
    }
}

// Called by osr_conv2d_fwd when in_dtype == OSR_F32 (generic argument checks already done there).
osr_status osr_conv_f32_run(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* residual, void* out,
                            hipStream_t st) {
    OSR_REQUIRE(p->out_dtype == OSR_F32, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd(f32): out_dtype must be f32");
    OSR_REQUIRE(p->cin >= 16 && p->cin % 16 == 0, OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd(f32): cin must be a multiple of 16, got %d", p->cin);
    OSR_REQUIRE(p->in_stride_w % 4 == 0 && p->in_stride_h % 4 == 0 && p->in_stride_n % 4 == 0, OSR_ERR_INVALID_ARG,
                "osr_conv2d_fwd(f32): input strides must be multiples of 4 elements (16-byte loads)");
    ConvF32Args a;
    a.p = *p; a.in = (const float*)in; a.w = (const float*)weight; a.bias = bias; a.res = (const float*)residual; a.out = (float*)out;
    a.M = (long long)p->n * p->ho * p->wo;
    const long long K = (long long)p->kh * p->kw * p->cin;
    OSR_REQUIRE(K <= (1ll << 30), OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd(f32): K too large");
    a.K = (int)K;
    a.tiles_n = (p->cout + 63) / 64;
    const long long tiles = (a.M + 63) / 64 * a.tiles_n;
    OSR_REQUIRE(tiles > 0 && tiles < (1ll << 31), OSR_ERR_UNSUPPORTED, "osr_conv2d_fwd(f32): grid too large");
    hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, st, a);
    OSR_CHECK_LAUNCH("osr_conv2d_fwd(f32)");
    return OSR_OK;
}
