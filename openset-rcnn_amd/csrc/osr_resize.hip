// On-device ResizeShortestEdge for uint8 images (include/osr.h: osr_resize_bilinear_u8): what [d2] ResizeTransform.apply_image does
// with PIL -- Image.resize(BILINEAR) = Pillow's separable triangle-filter resampling (src/libImaging/Resample.c): a horizontal
// pass into an 8-bit intermediate, then a vertical pass, 22-bit fixed-point coefficients, round-half-up, clip to [0, 255].
// (INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST of /root/reference/configs/Base-RCNN-FPN.yaml:43 via the loader built at train.py:129.)
// The coefficient tables (Pillow's precompute_coeffs + normalize_coeffs_8bpc) depend only on the two sizes and are computed by the
// caller on the host (host/data.py: pil_resample_coeffs); the kernels reproduce Pillow's integer arithmetic exactly, so the output is
// the PIL image bit for bit. HBM-trivial work (a 600 x 1000 frame is 1.8 MB): one thread per output pixel, three channels each.
#include "osr_common.h"

#define RS_PRECISION_BITS 22  // Pillow: 32 - 8 - 2

__device__ __forceinline__ unsigned char rs_clip8(int v) {
    v >>= RS_PRECISION_BITS;  // arithmetic shift, as Pillow's clip8 (lookup of in >> PRECISION_BITS, clamped to 0..255)
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// rows [y_first, y_first + rows) of the input -> tmp (rows, nw, 3): out(x) = sum_k in(xmin + k) * coef[x][k]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ in, long long in_row_stride, int y_first, int rows, int nw,
                                                       const int* __restrict__ xbounds, const int* __restrict__ xcoef, int kx, unsigned char* __restrict__ tmp) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * nw) return;
    const int xx = (int)(i % nw), r = (int)(i / nw);
    const int xmin = xbounds[2 * xx], xmax = xbounds[2 * xx + 1];
    const unsigned char* p = in + (long long)(y_first + r) * in_row_stride + (long long)xmin * 3;
    const int* k = xcoef + (long long)xx * kx;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int c = k[x];
        s0 += (int)p[3 * x] * c; s1 += (int)p[3 * x + 1] * c; s2 += (int)p[3 * x + 2] * c;
    }
    unsigned char* o = tmp + i * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
}

// tmp (rows, nw, 3) -> out (3, nh, nw): out(y) = sum_k tmp(ymin + k - y_first) * coef[y][k]
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ tmp, int y_first, int nh, int nw, const int* __restrict__ ybounds,
                                                       const int* __restrict__ ycoef, int ky, unsigned char* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nh * nw) return;
    const int xx = (int)(i % nw), yy = (int)(i / nw);
    const int ymin = ybounds[2 * yy], ymax = ybounds[2 * yy + 1];
    const int* k = ycoef + (long long)yy * ky;
    const unsigned char* p = tmp + ((long long)(ymin - y_first) * nw + xx) * 3;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) {
        const int c = k[y];
        const unsigned char* q = p + (long long)y * nw * 3;
        s0 += (int)q[0] * c; s1 += (int)q[1] * c; s2 += (int)q[2] * c;
    }
    const long long plane = (long long)nh * nw;
    out[i] = rs_clip8(s0); out[plane + i] = rs_clip8(s1); out[2 * plane + i] = rs_clip8(s2);
}

extern "C" int64_t osr_resize_tmp_bytes(int32_t h, int32_t nw) { return h > 0 && nw > 0 ? (int64_t)h * nw * 3 : 0; }

extern "C" osr_status osr_resize_bilinear_u8(const uint8_t* in, int32_t h, int32_t w, int64_t in_row_stride, const int32_t* xbounds, const int32_t* xcoef,
                                             int32_t kx, const int32_t* ybounds, const int32_t* ycoef, int32_t ky, int32_t y_first, int32_t y_rows,
                                             int32_t nh, int32_t nw, uint8_t* tmp, int64_t tmp_bytes, uint8_t* out, void* stream) {
    OSR_REQUIRE(in && xbounds && xcoef && ybounds && ycoef && tmp && out, OSR_ERR_INVALID_ARG, "osr_resize_bilinear_u8: null pointer");
    OSR_REQUIRE(h >= 1 && w >= 1 && nh >= 1 && nw >= 1 && kx >= 1 && ky >= 1 && in_row_stride >= (int64_t)w * 3, OSR_ERR_INVALID_ARG,
                "osr_resize_bilinear_u8: bad geometry");
    OSR_REQUIRE(y_first >= 0 && y_rows >= 1 && y_first + y_rows <= h, OSR_ERR_INVALID_ARG,
                "osr_resize_bilinear_u8: the rows the vertical pass reads, [y_first, y_first + y_rows), must lie inside the image");
    OSR_REQUIRE(tmp_bytes >= (int64_t)y_rows * nw * 3, OSR_ERR_INVALID_ARG, "osr_resize_bilinear_u8: tmp needs y_rows * nw * 3 bytes");
    hipStream_t st = (hipStream_t)stream;
    const long long n1 = (long long)y_rows * nw, n2 = (long long)nh * nw;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, in, (long long)in_row_stride, y_first, y_rows, nw, xbounds, xcoef, kx, tmp);
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, tmp, y_first, nh, nw, ybounds, ycoef, ky, out);
    OSR_CHECK_LAUNCH("osr_resize_bilinear_u8");
    return OSR_OK;
}
