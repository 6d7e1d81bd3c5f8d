// Sparse backward of the CF-RPN head's shared 3x3 convolution (training step).
//
// ClsFreeRPN.losses (/root/reference/openset_rcnn/modeling/proposal_generator/classification_free_rpn.py:446-490) sums its two terms
// over the SAMPLED anchors only (at most BATCH_SIZE_PER_IMAGE regression labels + BATCH_SIZE_PER_IMAGE objectness labels per
// image, :299-316), so the gradient of the head's five outputs -- and with it the gradient of the hidden state
// t = relu(conv3x3(p_l)) (ClsFreeRPNHead.forward, :159-161) -- is exactly zero on all but <= 2 * 256 * n of the ~1.4 M anchor rows
// of a 16-image batch. Autograd (and rounds 1-3 of this trainer) still ran the conv's data and weight gradients over the dense,
// almost-all-zero tensor: 2 x 1.7 TFLOP per step. The three launches here restate them on the non-zero rows:
//
//   osr_rpn_sparse_rows   the non-zero rows of d_out5 in ascending order (a fixed order: every later sum is reproducible), and the
//                         inverse map row -> list slot;
//   osr_rpn_gather_cols   the im2col row (9 taps x 256 channels, zero outside the map) of every listed anchor from the pyramid
//                         level it lives on, and its five output gradients -- the A operand of three small GEMMs on the existing
//                         MFMA kernels: the hidden state of the listed rows (recomputed: the forward need not store the 0.7 GB of
//                         it), the weight gradient dW = dt^T . cols, and the per-tap data gradient y = dt . W;
//   osr_rpn_scatter_cols_add   pixel-centric col2im: every pixel of the pyramid adds the (<= 9) rows of y that reach it, in tap
//                         order, in fp32, to the gradient already there (the RoI heads' share) and rounds once -- no atomics.
//
// Zero rows contribute exact zeros to the dense sums, so the results differ from the dense launches only in the order of the fp32
// additions.
#include "osr_common.h"

#define RS_BLOCKS 1024
#define RS_THREADS 256

__device__ __forceinline__ bool rs_row_nonzero(const float* __restrict__ d5, long long r) {
    const float* g = d5 + r * 5;
    return g[0] != 0.f || g[1] != 0.f || g[2] != 0.f || g[3] != 0.f || g[4] != 0.f;  // (NaN counts: it must reach the overflow check)
}

// rows [b * per, (b + 1) * per) belong to workgroup b, walked 256 at a time in ascending order
__global__ __launch_bounds__(RS_THREADS) void rs_count_kernel(const float* __restrict__ d5, long long rows, long long per, int* __restrict__ block_counts) {
    __shared__ int s_c[RS_THREADS / 64];
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    int cnt = 0;
    for (long long r = r0 + threadIdx.x; r < r1; r += RS_THREADS) cnt += rs_row_nonzero(d5, r) ? 1 : 0;
    cnt = (int)osr_wave_sum((float)cnt);  // (< 2^24: exact)
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ __launch_bounds__(RS_THREADS) void rs_fill_kernel(const float* __restrict__ d5, long long rows, long long per, const int* __restrict__ block_counts,
                                                            int cap, int* __restrict__ row_ids, int* __restrict__ row_map, int* __restrict__ count2) {
    __shared__ int s_w[RS_THREADS / 64], s_pre[RS_THREADS / 64], s_tot[RS_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // prefix of the workgroups before this one and the total, from the 1024 counts (4 per thread)
    int pre = 0, tot = 0;
    for (int i = threadIdx.x; i < RS_BLOCKS; i += RS_THREADS) {
        const int c = block_counts[i];
        tot += c;
        pre += i < (int)blockIdx.x ? c : 0;
    }
    pre = (int)osr_wave_sum((float)pre);
    tot = (int)osr_wave_sum((float)tot);
    if (lane == 0) { s_pre[wid] = pre; s_tot[wid] = tot; }
    __syncthreads();
    int base = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
    const int total = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { count2[0] = total < cap ? total : cap; count2[1] = total; }
    for (int j = total + (int)(blockIdx.x * RS_THREADS + threadIdx.x); j < cap; j += RS_BLOCKS * RS_THREADS) row_ids[j] = -1;  // the list's tail
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (long long rb = r0; rb < r1; rb += RS_THREADS) {
        const long long r = rb + threadIdx.x;
        const bool hit = r < r1 && rs_row_nonzero(d5, r);
        const unsigned long long bal = __ballot(hit);
        __syncthreads();  // (s_w of the previous turn has been read)
        if (lane == 0) s_w[wid] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wid; ++w) off += s_w[w];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (r < r1) {
            const bool take = hit && pos < cap;
            row_map[r] = take ? pos : -1;
            if (take) row_ids[pos] = (int)r;
        }
        base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
}

extern "C" int64_t osr_rpn_sparse_rows_workspace_bytes(void) { return (int64_t)RS_BLOCKS * 4; }

extern "C" osr_status osr_rpn_sparse_rows(const float* d_out5, int64_t rows, int32_t cap, int32_t* row_ids, int32_t* row_map, int32_t* count2,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(d_out5 && row_ids && row_map && count2 && workspace, OSR_ERR_INVALID_ARG, "osr_rpn_sparse_rows: null pointer");
    OSR_REQUIRE(rows >= 1 && rows < (1ll << 31) && cap >= 1, OSR_ERR_INVALID_ARG, "osr_rpn_sparse_rows: bad rows / cap");
    OSR_REQUIRE(workspace_bytes >= osr_rpn_sparse_rows_workspace_bytes(), OSR_ERR_WORKSPACE, "osr_rpn_sparse_rows: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    long long per = (rows + RS_BLOCKS - 1) / RS_BLOCKS;
    per = (per + RS_THREADS - 1) / RS_THREADS * RS_THREADS;
    hipLaunchKernelGGL(rs_count_kernel, dim3(RS_BLOCKS), dim3(RS_THREADS), 0, st, d_out5, (long long)rows, per, (int*)workspace);
    OSR_CHECK_LAUNCH("osr_rpn_sparse_rows(count)");
    hipLaunchKernelGGL(rs_fill_kernel, dim3(RS_BLOCKS), dim3(RS_THREADS), 0, st, d_out5, (long long)rows, per, (const int*)workspace, cap, row_ids, row_map, count2);
    OSR_CHECK_LAUNCH("osr_rpn_sparse_rows(fill)");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
struct RsGeom {
    int nl, n;
    int h[OSR_MAX_LEVELS], w[OSR_MAX_LEVELS];
    long long off[OSR_MAX_LEVELS + 1];  // first row of level l in the level-major row space; off[nl] = number of rows
    const void* feat[OSR_MAX_LEVELS];   // (n, h, w, 256) 2-byte elements
    void* grad[OSR_MAX_LEVELS];
};

__device__ __forceinline__ void rs_decode(const RsGeom& g, long long r, int& l, int& img, int& y, int& x) {
    l = 0;
#pragma unroll
    for (int i = 1; i < OSR_MAX_LEVELS; ++i) l += (i < g.nl && r >= g.off[i]) ? 1 : 0;
    const int local = (int)(r - g.off[l]);
    const int hw = g.h[l] * g.w[l];
    img = local / hw;
    const int rem = local - img * hw;
    y = rem / g.w[l];
    x = rem - y * g.w[l];
}

// one wave per (list slot, tap): 64 lanes x 8 bytes = the 256 channels of one pixel
__global__ __launch_bounds__(256) void rs_gather_kernel(RsGeom g, const int* __restrict__ row_ids, int cap, const float* __restrict__ d5,
                                                        uint2* __restrict__ cols, float* __restrict__ d5c) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= (long long)cap * 9) return;
    const int j = (int)(gw / 9), tap = (int)(gw - (long long)j * 9);
    const int rid = row_ids[j];
    uint2 v = make_uint2(0u, 0u);
    if (rid >= 0) {
        int l, img, y, x;
        rs_decode(g, rid, l, img, y, x);
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy >= 0 && yy < g.h[l] && xx >= 0 && xx < g.w[l])
            v = reinterpret_cast<const uint2*>(g.feat[l])[(((long long)img * g.h[l] + yy) * g.w[l] + xx) * 64 + lane];
    }
    cols[gw * 64 + lane] = v;
    if (tap == 0 && lane < 5) d5c[(long long)j * 5 + lane] = rid >= 0 ? d5[(long long)rid * 5 + lane] : 0.f;
}

extern "C" osr_status osr_rpn_gather_cols(const osr_rpn_levels* lv, const osr_pyramid* feats, int32_t feat_dtype, int32_t n, const int32_t* row_ids,
                                          int32_t cap, const float* d_out5, void* cols, float* d_out5_rows, void* stream) {
    OSR_REQUIRE(lv && feats && row_ids && d_out5 && cols && d_out5_rows, OSR_ERR_INVALID_ARG, "osr_rpn_gather_cols: null pointer");
    OSR_REQUIRE(feat_dtype == OSR_F16 || feat_dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_rpn_gather_cols: 2-byte features only");
    OSR_REQUIRE(lv->num_levels >= 1 && lv->num_levels <= OSR_MAX_LEVELS && lv->num_anchors == 1 && feats->num_levels == lv->num_levels && feats->c == 256,
                OSR_ERR_UNSUPPORTED, "osr_rpn_gather_cols: one anchor per location, 256 channels, the same levels in both descriptions");
    OSR_REQUIRE(n >= 1 && cap >= 1, OSR_ERR_INVALID_ARG, "osr_rpn_gather_cols: bad n / cap");
    RsGeom g{};
    g.nl = lv->num_levels; g.n = n;
    for (int l = 0; l < g.nl; ++l) {
        OSR_REQUIRE(feats->h[l] == lv->h[l] && feats->w[l] == lv->w[l] && feats->data[l], OSR_ERR_INVALID_ARG, "osr_rpn_gather_cols: level %d disagrees", l);
        OSR_REQUIRE(lv->offset[l] == (l == 0 ? 0 : lv->offset[l - 1] + (int64_t)n * lv->h[l - 1] * lv->w[l - 1]), OSR_ERR_INVALID_ARG,
                    "osr_rpn_gather_cols: levels must be level-major and dense");
        g.h[l] = lv->h[l]; g.w[l] = lv->w[l]; g.off[l] = lv->offset[l]; g.feat[l] = feats->data[l];
    }
    g.off[g.nl] = g.off[g.nl - 1] + (long long)n * g.h[g.nl - 1] * g.w[g.nl - 1];
    const long long waves = (long long)cap * 9;
    hipLaunchKernelGGL(rs_gather_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, row_ids, cap, d_out5, (uint2*)cols, d_out5_rows);
    OSR_CHECK_LAUNCH("osr_rpn_gather_cols");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// y: (cap, 9, 256) fp32, y[j][tap] = dt[j] . W[:, tap, :]. Pixel q of a level receives y[map[q - (tap offset)]][tap] for each tap whose
// source anchor is listed. A lane looks at one pixel's nine neighbours in the map; the wave then adds the hits one pixel at a time.
template <class T>
__global__ __launch_bounds__(256) void rs_scatter_kernel(RsGeom g, const int* __restrict__ row_map, const float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long long total = g.off[g.nl];
    const long long r = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    int mj[9];  // list slot of the anchor that reaches this pixel through tap t, or -1
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) mj[tap] = -1;
    int l = 0;
    if (r < total) {
        int img, py, px;
        rs_decode(g, r, l, img, py, px);
        const long long lb = g.off[l] + (long long)img * g.h[l] * g.w[l];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int qy = py - (tap / 3 - 1), qx = px - (tap % 3 - 1);
            if (qy >= 0 && qy < g.h[l] && qx >= 0 && qx < g.w[l]) mj[tap] = row_map[lb + (long long)qy * g.w[l] + qx];
        }
    }
    bool any = false;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) any |= mj[tap] >= 0;
    unsigned long long bal = __ballot(any);
    while (bal) {  // one reached pixel per turn: its nine slots come from the lane that looked at it, the nine rows of y are loaded side by side
        const int src = __builtin_ctzll(bal);
        bal &= bal - 1;
        const int sl = __shfl(l, src);
        const long long rr = r - lane + src;
        T* dst = reinterpret_cast<T*>(g.grad[sl]) + ((rr - g.off[sl]) * 256 + lane * 4);
        int j[9];
        float4 v[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            j[tap] = __shfl(mj[tap], src);
            v[tap] = *reinterpret_cast<const float4*>(y + ((long long)(j[tap] < 0 ? 0 : j[tap]) * 9 + tap) * 256 + lane * 4);
        }
        typedef T v4t __attribute__((ext_vector_type(4)));
        const v4t cur = *reinterpret_cast<const v4t*>(dst);
        float acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (float)cur[e];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (j[tap] < 0) continue;  // (wave-uniform; slot 0's row was loaded in its place and is dropped here)
            acc[0] += v[tap].x; acc[1] += v[tap].y; acc[2] += v[tap].z; acc[3] += v[tap].w;
        }
        v4t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = osr_from_float<T>(acc[e]);
        *reinterpret_cast<v4t*>(dst) = o;
    }
}

extern "C" osr_status osr_rpn_scatter_cols_add(const osr_rpn_levels* lv, int32_t n, const int32_t* row_map, const float* y, void* const* grads,
                                               int32_t grad_dtype, void* stream) {
    OSR_REQUIRE(lv && row_map && y && grads, OSR_ERR_INVALID_ARG, "osr_rpn_scatter_cols_add: null pointer");
    OSR_REQUIRE(grad_dtype == OSR_F16 || grad_dtype == OSR_BF16, OSR_ERR_UNSUPPORTED, "osr_rpn_scatter_cols_add: 2-byte gradients only");
    OSR_REQUIRE(lv->num_levels >= 1 && lv->num_levels <= OSR_MAX_LEVELS && lv->num_anchors == 1 && n >= 1, OSR_ERR_UNSUPPORTED,
                "osr_rpn_scatter_cols_add: one anchor per location");
    RsGeom g{};
    g.nl = lv->num_levels; g.n = n;
    for (int l = 0; l < g.nl; ++l) {
        OSR_REQUIRE(grads[l], OSR_ERR_INVALID_ARG, "osr_rpn_scatter_cols_add: level %d has no gradient tensor", l);
        OSR_REQUIRE(lv->offset[l] == (l == 0 ? 0 : lv->offset[l - 1] + (int64_t)n * lv->h[l - 1] * lv->w[l - 1]), OSR_ERR_INVALID_ARG,
                    "osr_rpn_scatter_cols_add: levels must be level-major and dense");
        g.h[l] = lv->h[l]; g.w[l] = lv->w[l]; g.off[l] = lv->offset[l]; g.grad[l] = grads[l];
    }
    g.off[g.nl] = g.off[g.nl - 1] + (long long)n * g.h[g.nl - 1] * g.w[g.nl - 1];
    const long long blocks = (g.off[g.nl] + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (grad_dtype == OSR_F16) hipLaunchKernelGGL(rs_scatter_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, st, g, row_map, y);
    else hipLaunchKernelGGL(rs_scatter_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, g, row_map, y);
    OSR_CHECK_LAUNCH("osr_rpn_scatter_cols_add");
    return OSR_OK;
}
