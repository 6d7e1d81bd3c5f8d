// Pre-processing, stem max-pool and p6 subsample for gfx950 (include/osr.h).
//   osr_preprocess   <- [d2] GeneralizedRCNN.preprocess_image + ImageList.from_tensors (train.py:135 entry)
//   osr_maxpool3x3s2 <- [d2] BasicStem: F.max_pool2d(kernel_size=3, stride=2, padding=1)
//   osr_subsample2   <- [d2] LastLevelMaxPool: F.max_pool2d(p5, kernel_size=1, stride=2)
// All three are pure HBM streams: one 8- or 16-byte vector per lane, coalesced along the fastest axis.
#include "osr_common.h"

template <class TS, class TD>
__global__ __launch_bounds__(256) void preprocess_kernel(const TS* __restrict__ src, int n, int h, int w, int hd, int wd,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         TD* __restrict__ dst) {
    const long long total = (long long)n * hd * wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % wd) - 3;
        const int y = (int)((i / wd) % hd) - 3;
        const int b = (int)(i / ((long long)wd * hd));
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const long long o = ((long long)b * 3 * h + y) * w + x;
            const long long plane = (long long)h * w;
            v0 = ((float)src[o] - m0) / s0;
            v1 = ((float)src[o + plane] - m1) / s1;
            v2 = ((float)src[o + 2 * plane] - m2) / s2;
        }
        typedef TD d4 __attribute__((ext_vector_type(4)));
        d4 o4 = {osr_from_float<TD>(v0), osr_from_float<TD>(v1), osr_from_float<TD>(v2), osr_from_float<TD>(0.f)};
        *reinterpret_cast<d4*>(dst + i * 4) = o4;
    }
}

extern "C" int32_t osr_stem_padded_width(int32_t wp) { return (wp + 8 + 7) / 8 * 8; }

extern "C" osr_status osr_preprocess(const void* src, int32_t src_is_u8, int32_t n, int32_t h, int32_t w, int32_t hp, int32_t wp,
                                     const float mean[3], const float stdv[3], void* dst, int32_t dst_dtype, void* stream) {
    OSR_REQUIRE(src && dst && mean && stdv, OSR_ERR_INVALID_ARG, "osr_preprocess: null pointer");
    OSR_REQUIRE(n >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w, OSR_ERR_INVALID_ARG, "osr_preprocess: bad sizes");
    OSR_REQUIRE(osr_dtype_ok(dst_dtype), OSR_ERR_UNSUPPORTED, "osr_preprocess: dst must be f16/bf16/f32");
    const int hd = hp + 6, wd = osr_stem_padded_width(wp);
    const long long total = (long long)n * hd * wd;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = (hipStream_t)stream;
#define OSR_PP(TS, TD) hipLaunchKernelGGL((preprocess_kernel<TS, TD>), dim3((unsigned)blocks), dim3(256), 0, st, (const TS*)src, n, h, w, hd, wd, \
                                          mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], (TD*)dst)
    if (src_is_u8) { if (dst_dtype == OSR_F16) OSR_PP(unsigned char, f16_t); else if (dst_dtype == OSR_BF16) OSR_PP(unsigned char, bf16_t); else OSR_PP(unsigned char, float); }
    else           { if (dst_dtype == OSR_F16) OSR_PP(float, f16_t); else if (dst_dtype == OSR_BF16) OSR_PP(float, bf16_t); else OSR_PP(float, float); }
#undef OSR_PP
    OSR_CHECK_LAUNCH("osr_preprocess");
    return OSR_OK;
}

// ---- 3x3/s2/p1 max pool, NHWC, 8 channels (16 B) per lane ----
template <class T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, int n, int hi, int wi, int c, int ho, int wo, T* __restrict__ out) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    const int c8 = c / 8;
    const long long total = (long long)n * ho * wo * c8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8) * 8;
        const int ox = (int)((i / c8) % wo), oy = (int)((i / ((long long)c8 * wo)) % ho), b = (int)(i / ((long long)c8 * wo * ho));
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -3.402823466e38f;
        for (int dy = 0; dy < 3; ++dy) {
            const int y = oy * 2 - 1 + dy;
            if (y < 0 || y >= hi) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int x = ox * 2 - 1 + dx;
                if (x < 0 || x >= wi) continue;
                const t8 v = *reinterpret_cast<const t8*>(in + (((long long)b * hi + y) * wi + x) * c + cc);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], (float)v[k]);
            }
        }
        t8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (T)m[k];
        *reinterpret_cast<t8*>(out + (((long long)b * ho + oy) * wo + ox) * c + cc) = o;
    }
}

extern "C" osr_status osr_maxpool3x3s2(const void* in, int32_t n, int32_t hi, int32_t wi, int32_t c, void* out, int32_t dtype, void* stream) {
    OSR_REQUIRE(in && out, OSR_ERR_INVALID_ARG, "osr_maxpool3x3s2: null pointer");
    OSR_REQUIRE(n >= 1 && hi >= 1 && wi >= 1 && c >= 8 && c % 8 == 0, OSR_ERR_INVALID_ARG, "osr_maxpool3x3s2: bad sizes (c %% 8 == 0)");
    OSR_REQUIRE(osr_dtype_ok(dtype), OSR_ERR_UNSUPPORTED, "osr_maxpool3x3s2: bad dtype");
    const int ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OSR_F16) hipLaunchKernelGGL(maxpool_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16_t*)in, n, hi, wi, c, ho, wo, (f16_t*)out);
    else if (dtype == OSR_F32) hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)in, n, hi, wi, c, ho, wo, (float*)out);
    else hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)in, n, hi, wi, c, ho, wo, (bf16_t*)out);
    OSR_CHECK_LAUNCH("osr_maxpool3x3s2");
    return OSR_OK;
}

// ---- stride-2 subsample (p6), NHWC, 16 B per lane; element-size agnostic (c8 = 16-byte chunks per pixel) ----
__global__ __launch_bounds__(256) void subsample_kernel(const uint4* __restrict__ in, int n, int hi, int wi, int c8, int ho, int wo, uint4* __restrict__ out) {
    const long long total = (long long)n * ho * wo * c8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8);
        const int ox = (int)((i / c8) % wo), oy = (int)((i / ((long long)c8 * wo)) % ho), b = (int)(i / ((long long)c8 * wo * ho));
        out[i] = in[(((long long)b * hi + oy * 2) * wi + ox * 2) * c8 + cc];
    }
}

extern "C" osr_status osr_subsample2(const void* in, int32_t n, int32_t hi, int32_t wi, int32_t c, void* out, int32_t dtype, void* stream) {
    OSR_REQUIRE(in && out, OSR_ERR_INVALID_ARG, "osr_subsample2: null pointer");
    OSR_REQUIRE(n >= 1 && hi >= 1 && wi >= 1 && c >= 8 && c % 8 == 0, OSR_ERR_INVALID_ARG, "osr_subsample2: bad sizes (c %% 8 == 0)");
    OSR_REQUIRE(osr_dtype_ok(dtype), OSR_ERR_UNSUPPORTED, "osr_subsample2: bad dtype");
    const int ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    const int c8 = c * osr_dtype_size(dtype) / 16;  // 16-byte chunks per pixel
    const long long total = (long long)n * ho * wo * c8;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, n, hi, wi, c8, ho, wo, (uint4*)out);
    OSR_CHECK_LAUNCH("osr_subsample2");
    return OSR_OK;
}
