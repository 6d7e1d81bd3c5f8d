// Second-stage tail kernels for gfx950 (include/osr.h): box predictor tail, segmented stable sort + greedy
// per-class NMS + top-k, row gather, L2 normalise, PLN tail, softmax-classifier candidates, final assembly.
//
// Reference call sites (/root/reference/openset_rcnn/modeling/roi_heads/):
//   osrcnn_fast_rcnn.py:89-145,248-264,403-450   prototype_learning_network.py:189-230
//   softmax_classifier.py:47-168,287-346
// Compile with -ffp-contract=off: IoU / clip / threshold arithmetic must round like the oracle's so that
// kept-index lists are bit-exact.
#include "osr_common.h"
#include "osr_pln_dist.h"

#define SCALE_CLAMP_F 4.135166556742356f  // log(1000/16)

// ------------------------------------------------------------------------------------------------------
// box predictor tail: wave per RoI row
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_pred_tail_kernel(const float* __restrict__ x, long long m, int k,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ proposals, const float* __restrict__ ctr,
                                                            const int* __restrict__ batch_idx, const int* __restrict__ image_hw,
                                                            float wx, float wy, float ww, float wh, int mean_type, float score_thresh,
                                                            float* __restrict__ pred_deltas, float* __restrict__ pred_iou,
                                                            float* __restrict__ boxes, float* __restrict__ score, int* __restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [5][k]
    for (int i = threadIdx.x; i < 5 * k; i += blockDim.x) s_w[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (long long r = (long long)blockIdx.x * nw + wid; r < m; r += (long long)gridDim.x * nw) {
        const int bi = batch_idx[r];
        if (bi < 0) {
            if (lane == 0) {
                *reinterpret_cast<float4*>(pred_deltas + r * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(boxes + r * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                pred_iou[r] = 0.f; score[r] = 0.f; cand[r] = 0;
            }
            continue;
        }
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f;
        const float* row = x + r * k;
        for (int c = lane * 4; c < k; c += 256) {
            const float4 q = *reinterpret_cast<const float4*>(row + c);
            const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d0 += v[j] * s_w[c + j];
                d1 += v[j] * s_w[k + c + j];
                d2 += v[j] * s_w[2 * k + c + j];
                d3 += v[j] * s_w[3 * k + c + j];
                d4 += v[j] * s_w[4 * k + c + j];
            }
        }
        d0 = osr_wave_sum(d0); d1 = osr_wave_sum(d1); d2 = osr_wave_sum(d2); d3 = osr_wave_sum(d3); d4 = osr_wave_sum(d4);
        if (lane == 0) {
            d0 += b[0]; d1 += b[1]; d2 += b[2]; d3 += b[3];
            const float iou = 1.0f / (1.0f + expf(-(d4 + b[4])));
            const float4 p = *reinterpret_cast<const float4*>(proposals + r * 4);
            // [d2] Box2BoxTransform.apply_deltas
            const float pw_ = p.z - p.x, ph_ = p.w - p.y;
            const float cx = p.x + 0.5f * pw_, cy = p.y + 0.5f * ph_;
            const float dx = d0 / wx, dy = d1 / wy;
            float dw = d2 / ww, dh = d3 / wh;
            dw = dw > SCALE_CLAMP_F ? SCALE_CLAMP_F : dw;  // NaN stays NaN like torch.clamp(max=)
            dh = dh > SCALE_CLAMP_F ? SCALE_CLAMP_F : dh;
            const float pcx = dx * pw_ + cx, pcy = dy * ph_ + cy;
            const float qw = expf(dw) * pw_, qh = expf(dh) * ph_;
            float x1 = pcx - 0.5f * qw, y1 = pcy - 0.5f * qh, x2 = pcx + 0.5f * qw, y2 = pcy + 0.5f * qh;
            const float c_ = ctr[r];
            const float s = mean_type == 0 ? sqrtf(iou * c_) : (iou + c_) / 2.0f;
            const bool valid = osr_finite(x1) && osr_finite(y1) && osr_finite(x2) && osr_finite(y2) && osr_finite(s);
            const float ih = (float)image_hw[bi * 2], iw = (float)image_hw[bi * 2 + 1];
            x1 = fminf(fmaxf(x1, 0.f), iw); y1 = fminf(fmaxf(y1, 0.f), ih);
            x2 = fminf(fmaxf(x2, 0.f), iw); y2 = fminf(fmaxf(y2, 0.f), ih);
            *reinterpret_cast<float4*>(pred_deltas + r * 4) = make_float4(d0, d1, d2, d3);
            pred_iou[r] = iou;
            *reinterpret_cast<float4*>(boxes + r * 4) = make_float4(x1, y1, x2, y2);
            score[r] = s;
            cand[r] = (valid && s > score_thresh) ? 1 : 0;
        }
    }
}

extern "C" osr_status osr_box_predictor_tail(const float* x, int64_t m, int32_t k, const float* w, const float* b,
                                             const float* proposals, const float* ctr, const int32_t* batch_idx,
                                             const int32_t* image_hw, const float reg_weights[4], int32_t mean_type,
                                             float score_thresh, float* pred_deltas, float* pred_iou, float* boxes,
                                             float* score, int32_t* cand, void* stream) {
    OSR_REQUIRE(x && w && b && proposals && ctr && batch_idx && image_hw && reg_weights && pred_deltas && pred_iou && boxes && score && cand,
                OSR_ERR_INVALID_ARG, "osr_box_predictor_tail: null pointer");
    OSR_REQUIRE(k > 0 && k % 4 == 0 && k <= 3072, OSR_ERR_UNSUPPORTED, "osr_box_predictor_tail: k must be a multiple of 4 and <= 3072, got %d", k);
    OSR_REQUIRE(m >= 0 && (mean_type == 0 || mean_type == 1), OSR_ERR_INVALID_ARG, "osr_box_predictor_tail: bad m / mean_type");
    if (m == 0) return OSR_OK;
    long long blocks = (m + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(box_pred_tail_kernel, dim3((unsigned)blocks), dim3(256), (size_t)5 * k * 4, (hipStream_t)stream, x, (long long)m, k, w, b,
                       proposals, ctr, batch_idx, image_hw, reg_weights[0], reg_weights[1], reg_weights[2], reg_weights[3], mean_type,
                       score_thresh, pred_deltas, pred_iou, boxes, score, cand);
    OSR_CHECK_LAUNCH("osr_box_predictor_tail");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// segmented stable sort (score desc, index asc) + greedy NMS + top-k
// ------------------------------------------------------------------------------------------------------
#define SORT_THREADS 1024
#define SORT_LDS_CAP 8192

template <class P>
__device__ __forceinline__ void bitonic_desc_any(P buf, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = buf[i], b = buf[ixj];
                    bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { buf[i] = b; buf[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

static inline long long pow2ceil(long long v) { long long p = 1; while (p < v) p <<= 1; return p; }

// workspace per segment: order[seg_stride] int32, then (16-byte aligned) keys[pow2ceil(seg_stride)] u64
__global__ __launch_bounds__(SORT_THREADS) void seg_sort_kernel(const float* __restrict__ scores, const int* __restrict__ cand,
                                                                long long seg_stride, const int* __restrict__ seg_len,
                                                                int* __restrict__ ws_order, int* __restrict__ ws_count,
                                                                unsigned long long* __restrict__ ws_keys, long long keys_stride, int lds_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];  // [lds_cap] keys, then 32 ints of scan scratch
    int* s_scan = reinterpret_cast<int*>(s_keys + lds_cap);
    const int seg = blockIdx.x, tid = threadIdx.x;
    long long len = seg_len[seg];
    if (len > seg_stride) len = seg_stride;
    if (len < 0) len = 0;
    const float* sc = scores + seg * seg_stride;
    const int* cd = cand ? cand + seg * seg_stride : nullptr;
    int* order = ws_order + seg * seg_stride;
    int np = 1;
    while (np < len) np <<= 1;
    const bool in_lds = np <= lds_cap;
    unsigned long long* gk = ws_keys + seg * keys_stride;
    int nvalid = 0;
    for (int i = tid; i < np; i += blockDim.x) {
        unsigned long long key = 0ull;
        if (i < len && (!cd || cd[i])) {
            key = ((unsigned long long)osr_float_key(sc[i]) << 32) | (unsigned int)(0xffffffffu - (unsigned int)i);
            ++nvalid;
        }
        if (in_lds) s_keys[i] = key; else gk[i] = key;
    }
    int tot;
    osr_block_excl_scan(nvalid, s_scan, &tot);
    __syncthreads();
    if (in_lds) bitonic_desc_any(s_keys, np); else bitonic_desc_any(gk, np);
    for (int i = tid; i < tot; i += blockDim.x) {
        unsigned long long key = in_lds ? s_keys[i] : gk[i];
        order[i] = (int)(0xffffffffu - (unsigned int)(key & 0xffffffffull));
    }
    if (tid == 0) ws_count[seg] = tot;
}

#define NMS_MAX_TOPK 1024

__global__ __launch_bounds__(64) void nms_greedy_kernel(const float* __restrict__ boxes, const int* __restrict__ cls, long long seg_stride,
                                                        const int* __restrict__ ws_order, const int* __restrict__ ws_count, float thr,
                                                        int topk, int* __restrict__ keep, int* __restrict__ keep_count) {
    __shared__ float4 s_box[NMS_MAX_TOPK];
    __shared__ float s_area[NMS_MAX_TOPK];
    __shared__ int s_cls[NMS_MAX_TOPK];
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int n = ws_count[seg];
    const int* order = ws_order + seg * seg_stride;
    const float* bx = boxes + seg * seg_stride * 4;
    const int* cl = cls ? cls + seg * seg_stride : nullptr;
    int* kp = keep + (long long)seg * topk;
    if (thr >= 1.0f) {  // IoU > thr can never hold: pure sort + top-k
        const int nk = n < topk ? n : topk;
        for (int i = lane; i < nk; i += 64) kp[i] = order[i];
        for (int i = nk + lane; i < topk; i += 64) kp[i] = -1;
        if (lane == 0) keep_count[seg] = nk;
        return;
    }
    int nkept = 0;
    for (int base = 0; base < n && nkept < topk; base += 64) {
        const int j = base + lane;
        bool alive = j < n;
        int idx = -1, c = 0;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        float area = 0.f;
        if (alive) {
            idx = order[j];
            q = *reinterpret_cast<const float4*>(bx + (long long)idx * 4);
            c = cl ? cl[idx] : 0;
            area = (q.z - q.x) * (q.w - q.y);
            for (int t = 0; t < nkept; ++t) {
                if (s_cls[t] != c) continue;
                const float4 kq = s_box[t];
                const float xx1 = fmaxf(kq.x, q.x), yy1 = fmaxf(kq.y, q.y), xx2 = fminf(kq.z, q.z), yy2 = fminf(kq.w, q.w);
                const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
                const float inter = w * h;
                const float ovr = inter / (s_area[t] + area - inter);
                if (ovr > thr) { alive = false; break; }
            }
        }
        unsigned long long msk = __ballot(alive);
        while (msk && nkept < topk) {
            const int i = __ffsll((long long)msk) - 1;
            const float ix1 = __shfl(q.x, i, 64), iy1 = __shfl(q.y, i, 64), ix2 = __shfl(q.z, i, 64), iy2 = __shfl(q.w, i, 64);
            const float ia = __shfl(area, i, 64);
            const int ic = __shfl(c, i, 64), iidx = __shfl(idx, i, 64);
            if (lane == 0) {
                s_box[nkept] = make_float4(ix1, iy1, ix2, iy2);
                s_area[nkept] = ia;
                s_cls[nkept] = ic;
                kp[nkept] = iidx;
            }
            ++nkept;
            if (lane == i) alive = false;
            if (alive && lane > i && c == ic) {
                const float xx1 = fmaxf(ix1, q.x), yy1 = fmaxf(iy1, q.y), xx2 = fminf(ix2, q.z), yy2 = fminf(iy2, q.w);
                const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
                const float inter = w * h;
                const float ovr = inter / (ia + area - inter);
                if (ovr > thr) alive = false;
            }
            msk = __ballot(alive && lane > i);
        }
        __syncthreads();  // single wave: orders lane 0's LDS writes before the next chunk's reads
    }
    for (int i = nkept + lane; i < topk; i += 64) kp[i] = -1;
    if (lane == 0) keep_count[seg] = nkept;
}

static void nms_ws_layout(int num_segments, long long seg_stride, long long* off_count, long long* off_keys, long long* keys_stride, long long* total) {
    long long order_bytes = (long long)num_segments * seg_stride * 4;
    long long oc = (order_bytes + 15) / 16 * 16;
    long long ok = oc + ((long long)num_segments * 4 + 15) / 16 * 16;
    long long ks = pow2ceil(seg_stride);
    *off_count = oc; *off_keys = ok; *keys_stride = ks;
    *total = ok + (long long)num_segments * ks * 8;
}

extern "C" int64_t osr_nms_topk_workspace_bytes(int32_t num_segments, int64_t seg_stride) {
    if (num_segments < 1 || seg_stride < 1 || seg_stride > (1ll << 24)) { osr_set_error("osr_nms_topk_workspace_bytes: bad arguments"); return OSR_ERR_INVALID_ARG; }
    long long a, b, c, t;
    nms_ws_layout(num_segments, seg_stride, &a, &b, &c, &t);
    return t;
}

extern "C" osr_status osr_nms_topk(const float* boxes, const float* scores, const int32_t* cls, const int32_t* cand,
                                   int32_t num_segments, int64_t seg_stride, const int32_t* seg_len, float thr, int32_t topk,
                                   int32_t* keep, int32_t* keep_count, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(boxes && scores && seg_len && keep && keep_count && workspace, OSR_ERR_INVALID_ARG, "osr_nms_topk: null pointer");
    OSR_REQUIRE(num_segments >= 1 && seg_stride >= 1 && seg_stride <= (1ll << 24), OSR_ERR_INVALID_ARG, "osr_nms_topk: bad segment geometry");
    OSR_REQUIRE(topk >= 1, OSR_ERR_INVALID_ARG, "osr_nms_topk: topk must be >= 1");
    OSR_REQUIRE(thr >= 1.0f || topk <= NMS_MAX_TOPK, OSR_ERR_UNSUPPORTED, "osr_nms_topk: topk > %d needs thr >= 1", NMS_MAX_TOPK);
    OSR_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)boxes & 15) == 0, OSR_ERR_INVALID_ARG, "osr_nms_topk: boxes/workspace must be 16-byte aligned");
    long long oc, ok, ks, total;
    nms_ws_layout(num_segments, seg_stride, &oc, &ok, &ks, &total);
    OSR_REQUIRE(workspace_bytes >= total, OSR_ERR_WORKSPACE, "osr_nms_topk: workspace %lld < %lld bytes", (long long)workspace_bytes, total);
    char* ws = (char*)workspace;
    int* ws_order = (int*)ws;
    int* ws_count = (int*)(ws + oc);
    unsigned long long* ws_keys = (unsigned long long*)(ws + ok);
    hipStream_t st = (hipStream_t)stream;
    long long np = pow2ceil(seg_stride);
    const int lds_cap = (int)(np <= SORT_LDS_CAP ? np : SORT_LDS_CAP);
    const size_t smem = (size_t)lds_cap * 8 + 128;
    static osr_dev_mask attr_set{0};
    osr_once_per_device(attr_set, [] {  // > 64 KB of dynamic LDS needs the opt-in (per device); 160 KB per CU on gfx950
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seg_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_CAP * 8 + 128);
    });
    hipLaunchKernelGGL(seg_sort_kernel, dim3(num_segments), dim3(SORT_THREADS), smem, st, scores, cand, (long long)seg_stride, seg_len, ws_order,
                       ws_count, ws_keys, ks, lds_cap);
    OSR_CHECK_LAUNCH("osr_nms_topk(sort)");
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(num_segments), dim3(64), 0, st, boxes, cls, (long long)seg_stride, ws_order, ws_count, thr, topk,
                       keep, keep_count);
    OSR_CHECK_LAUNCH("osr_nms_topk(nms)");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// gather rows
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, long long seg_stride, int row_elems,
                                                          const int* __restrict__ keep, const int* __restrict__ keep_count, int topk,
                                                          float* __restrict__ dst) {
    const int j = blockIdx.x, seg = blockIdx.y;
    float* d = dst + ((long long)seg * topk + j) * row_elems;
    const int cnt = keep_count[seg];
    const int idx = j < cnt ? keep[(long long)seg * topk + j] : -1;
    if (idx < 0) {
        for (int i = threadIdx.x; i < row_elems; i += blockDim.x) d[i] = 0.f;
        return;
    }
    const float* s = src + (seg * seg_stride + idx) * row_elems;
    if ((row_elems & 3) == 0) {
        for (int i = threadIdx.x * 4; i < row_elems; i += blockDim.x * 4)
            *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
    } else {
        for (int i = threadIdx.x; i < row_elems; i += blockDim.x) d[i] = s[i];
    }
}

extern "C" osr_status osr_gather_rows(const float* src, int64_t seg_stride, int32_t row_elems, const int32_t* keep,
                                      const int32_t* keep_count, int32_t num_segments, int32_t topk, float* dst, void* stream) {
    OSR_REQUIRE(src && keep && keep_count && dst, OSR_ERR_INVALID_ARG, "osr_gather_rows: null pointer");
    OSR_REQUIRE(num_segments >= 1 && num_segments <= 65535 && topk >= 1 && row_elems >= 1 && seg_stride >= 1, OSR_ERR_INVALID_ARG, "osr_gather_rows: bad sizes");
    OSR_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_gather_rows: src/dst must be 16-byte aligned");
    const int threads = row_elems >= 1024 ? 256 : 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(topk, num_segments), dim3(threads), 0, (hipStream_t)stream, src, (long long)seg_stride, row_elems,
                       keep, keep_count, topk, dst);
    OSR_CHECK_LAUNCH("osr_gather_rows");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// L2 normalise rows (F.normalize): x / max(||x||, 1e-12); wave per row
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int rows, int d, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    float ss = 0.f;
    for (int i = lane; i < d; i += 64) { float v = x[(long long)r * d + i]; ss += v * v; }
    ss = osr_wave_sum(ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    for (int i = lane; i < d; i += 64) out[(long long)r * d + i] = x[(long long)r * d + i] / den;
}

extern "C" osr_status osr_l2_normalize_rows(const float* x, int32_t rows, int32_t d, float* out, void* stream) {
    OSR_REQUIRE(x && out && rows >= 0 && d >= 1, OSR_ERR_INVALID_ARG, "osr_l2_normalize_rows: bad arguments");
    if (rows == 0) return OSR_OK;
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, d, out);
    OSR_CHECK_LAUNCH("osr_l2_normalize_rows");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// PLN tail: wave per embedding row
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pln_tail_kernel(const float* __restrict__ emb, long long rows, int d,
                                                       const float* __restrict__ protos, int num_known, int reps, int dist_type, float unk_thr,
                                                       long long unknown_id, const long long* __restrict__ class_map,
                                                       const int* __restrict__ rows_valid, int seg_rows,
                                                       long long* __restrict__ pred_class, float* __restrict__ min_dist) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];  // [num_known*reps][d]
    const int np = num_known * reps;
    for (int i = threadIdx.x; i < np * d; i += blockDim.x) s_p[i] = protos[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (long long r = (long long)blockIdx.x * nw + wid; r < rows; r += (long long)gridDim.x * nw) {
        if (rows_valid) {
            const long long seg = r / seg_rows;
            if ((int)(r - seg * seg_rows) >= rows_valid[seg]) {
                if (lane == 0) { pred_class[r] = -1; min_dist[r] = 0.f; }
                continue;
            }
        }
        const float* e = emb + r * d;
        float ss = 0.f;
        for (int i = lane; i < d; i += 64) { float v = e[i]; ss += v * v; }
        ss = osr_wave_sum(ss);
        const float den = fmaxf(sqrtf(ss), 1e-12f);
        float best = 0.f;
        int best_c = -1;
        for (int c = 0; c < num_known; ++c) {
            float md = 0.f;
            for (int q = 0; q < reps; ++q) {
                const float* p = s_p + (long long)(c * reps + q) * d;
                const float dist = osr_pln_distance([&](int i) { return e[i] / den; }, p, d, lane, dist_type);
                md = (q == 0 || dist < md) ? dist : md;
            }
            if (best_c < 0 || md < best) { best = md; best_c = c; }  // strict <: lower class wins ties
        }
        if (lane == 0) {
            long long cls = class_map ? class_map[best_c] : (long long)best_c;
            if (best > unk_thr) cls = unknown_id;
            pred_class[r] = cls;
            min_dist[r] = best;
        }
    }
}

extern "C" osr_status osr_pln_tail_ex(const float* emb, int64_t rows, int32_t d, const float* protos_normed, int32_t num_known,
                                      int32_t reps, int32_t distance_type, float unk_thr, int64_t unknown_id, const int64_t* class_map,
                                      const int32_t* rows_valid, int32_t seg_rows, int64_t* pred_class, float* min_dist, void* stream) {
    OSR_REQUIRE(distance_type >= OSR_DIST_COS && distance_type <= OSR_DIST_L2, OSR_ERR_INVALID_ARG, "osr_pln_tail: distance_type %d", distance_type);
    OSR_REQUIRE(emb && protos_normed && pred_class && min_dist, OSR_ERR_INVALID_ARG, "osr_pln_tail: null pointer");
    OSR_REQUIRE(rows >= 0 && d >= 1 && num_known >= 1 && reps >= 1, OSR_ERR_INVALID_ARG, "osr_pln_tail: bad sizes");
    OSR_REQUIRE((long long)num_known * reps * d <= 36864, OSR_ERR_UNSUPPORTED, "osr_pln_tail: prototypes exceed the 144 KB LDS table");
    OSR_REQUIRE(!rows_valid || seg_rows >= 1, OSR_ERR_INVALID_ARG, "osr_pln_tail: seg_rows must be >= 1 with rows_valid");
    if (rows == 0) return OSR_OK;
    long long blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    size_t smem = (size_t)num_known * reps * d * 4;
    if (smem > 64 * 1024) {
        static osr_dev_mask attr{0};
        osr_once_per_device(attr, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pln_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); });
    }
    hipLaunchKernelGGL(pln_tail_kernel, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, emb, (long long)rows, d, protos_normed,
                       num_known, reps, distance_type, unk_thr, (long long)unknown_id, (const long long*)class_map, rows_valid, seg_rows,
                       (long long*)pred_class, min_dist);
    OSR_CHECK_LAUNCH("osr_pln_tail");
    return OSR_OK;
}

extern "C" osr_status osr_pln_tail(const float* emb, int64_t rows, int32_t d, const float* protos_normed, int32_t num_known,
                                   int32_t reps, float unk_thr, int64_t unknown_id, const int64_t* class_map,
                                   const int32_t* rows_valid, int32_t seg_rows, int64_t* pred_class, float* min_dist, void* stream) {
    return osr_pln_tail_ex(emb, rows, d, protos_normed, num_known, reps, OSR_DIST_COS, unk_thr, unknown_id, class_map, rows_valid, seg_rows, pred_class,
                           min_dist, stream);
}

// ------------------------------------------------------------------------------------------------------
// softmax classifier candidates: block per image, thread per detection
// ------------------------------------------------------------------------------------------------------
#define SM_MAX_KNOWN 64

__global__ __launch_bounds__(1024) void softmax_cand_kernel(const float* __restrict__ logits, int K, const float* __restrict__ det_boxes,
                                                            const float* __restrict__ det_scores, const long long* __restrict__ pred_class,
                                                            const int* __restrict__ det_count, int seg_rows, long long unknown_id,
                                                            float known_thresh, float unknown_thresh, float* __restrict__ k_boxes,
                                                            float* __restrict__ k_scores, int* __restrict__ k_cls, int* __restrict__ k_det,
                                                            int* __restrict__ k_count, float* __restrict__ u_boxes, float* __restrict__ u_scores,
                                                            int* __restrict__ u_det, int* __restrict__ u_count) {
    __shared__ int s_scan[32];
    const int img = blockIdx.x, tid = threadIdx.x;
    int cnt = det_count[img];
    if (cnt > seg_rows) cnt = seg_rows;
    const long long kcap = (long long)seg_rows * K;
    int run_k = 0, run_u = 0;
    for (int j0 = 0; j0 < seg_rows; j0 += blockDim.x) {
        const int j = j0 + tid;
        const long long r = (long long)img * seg_rows + j;
        int nk = 0, nu = 0;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        float mx = 0.f, sum = 1.f;
        bool known = false, okbox = false;
        if (j < cnt) {
            bx = *reinterpret_cast<const float4*>(det_boxes + r * 4);
            okbox = osr_finite(bx.x) && osr_finite(bx.y) && osr_finite(bx.z) && osr_finite(bx.w);
            const long long pc = pred_class[r];
            known = pc != unknown_id;
            if (known) {
                const float* lg = logits + r * (K + 1);
                mx = lg[0];
                for (int c = 1; c <= K; ++c) mx = fmaxf(mx, lg[c]);
                sum = 0.f;
                for (int c = 0; c <= K; ++c) sum += expf(lg[c] - mx);
                bool okp = true;
                for (int c = 0; c <= K; ++c) okp = okp && osr_finite(expf(lg[c] - mx) / sum);
                if (okbox && okp)
                    for (int c = 0; c < K; ++c) nk += (expf(lg[c] - mx) / sum) > known_thresh;
            } else {
                const float s = det_scores[r];
                nu = (okbox && osr_finite(s) && s > unknown_thresh) ? 1 : 0;
            }
        }
        int tk, tu;
        const int pk = run_k + osr_block_excl_scan(nk, s_scan, &tk);
        const int pu = run_u + osr_block_excl_scan(nu, s_scan, &tu);
        if (nk) {
            const float* lg = logits + r * (K + 1);
            int o = pk;
            for (int c = 0; c < K; ++c) {
                const float p = expf(lg[c] - mx) / sum;
                if (p > known_thresh) {
                    const long long q = (long long)img * kcap + o;
                    *reinterpret_cast<float4*>(k_boxes + q * 4) = bx;
                    k_scores[q] = p; k_cls[q] = c; k_det[q] = j;
                    ++o;
                }
            }
        }
        if (nu) {
            const long long q = (long long)img * seg_rows + pu;
            *reinterpret_cast<float4*>(u_boxes + q * 4) = bx;
            u_scores[q] = det_scores[r]; u_det[q] = j;
        }
        run_k += tk;
        run_u += tu;
    }
    if (tid == 0) { k_count[img] = run_k; u_count[img] = run_u; }
}

extern "C" osr_status osr_softmax_candidates(const float* logits, int32_t num_known, const float* det_boxes, const float* det_scores,
                                             const int64_t* pred_class, const int32_t* det_count, int32_t n, int32_t seg_rows,
                                             int64_t unknown_id, float known_thresh, float unknown_thresh, float* k_boxes,
                                             float* k_scores, int32_t* k_cls, int32_t* k_det, int32_t* k_count, float* u_boxes,
                                             float* u_scores, int32_t* u_det, int32_t* u_count, void* stream) {
    OSR_REQUIRE(logits && det_boxes && det_scores && pred_class && det_count && k_boxes && k_scores && k_cls && k_det && k_count && u_boxes &&
                    u_scores && u_det && u_count, OSR_ERR_INVALID_ARG, "osr_softmax_candidates: null pointer");
    OSR_REQUIRE(n >= 1 && seg_rows >= 1 && num_known >= 1 && num_known <= SM_MAX_KNOWN, OSR_ERR_INVALID_ARG, "osr_softmax_candidates: bad sizes");
    OSR_REQUIRE((((uintptr_t)det_boxes | (uintptr_t)k_boxes | (uintptr_t)u_boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_softmax_candidates: box arrays must be 16-byte aligned");
    hipLaunchKernelGGL(softmax_cand_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, logits, num_known, det_boxes, det_scores,
                       (const long long*)pred_class, det_count, seg_rows, (long long)unknown_id, known_thresh, unknown_thresh, k_boxes, k_scores,
                       k_cls, k_det, k_count, u_boxes, u_scores, u_det, u_count);
    OSR_CHECK_LAUNCH("osr_softmax_candidates");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// [d2] FastRCNNOutputLayers.inference -> fast_rcnn_inference_single_image, everything before the NMS (BASELINE config 1:
// Base-RCNN-FPN.yaml's StandardROIHeads): softmax over K+1 logits, class-specific Box2BoxTransform decode (K*4 deltas per
// row; 4 when class-agnostic), rows with a non-finite box or probability dropped, clip, (row, class) pairs with
// p > score_thresh emitted in row-major order (= nonzero()). Block per image, thread per proposal row.
// ------------------------------------------------------------------------------------------------------
#define FR_MAX_CLASSES 128

__global__ __launch_bounds__(1024) void fastrcnn_cand_kernel(const float* __restrict__ logits, const float* __restrict__ deltas, int K, int kbox,
                                                             const float* __restrict__ prop_boxes, const int* __restrict__ prop_count,
                                                             int seg_rows, const int* __restrict__ image_hw, float4 rw, float score_thresh,
                                                             float* __restrict__ c_boxes, float* __restrict__ c_scores, int* __restrict__ c_cls,
                                                             int* __restrict__ c_row, int* __restrict__ c_count) {
    __shared__ int s_scan[32];
    const int img = blockIdx.x, tid = threadIdx.x;
    int cnt = prop_count[img];
    if (cnt > seg_rows) cnt = seg_rows;
    const long long ccap = (long long)seg_rows * K;
    const float ih = (float)image_hw[img * 2], iw = (float)image_hw[img * 2 + 1];
    const float kClamp = 4.135166556742356f;  // log(1000 / 16)
    int run = 0;
    for (int j0 = 0; j0 < seg_rows; j0 += blockDim.x) {
        const int j = j0 + tid;
        const long long r = (long long)img * seg_rows + j;
        int nk = 0;
        float mx = 0.f, sum = 1.f;
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < cnt) {
            pb = *reinterpret_cast<const float4*>(prop_boxes + r * 4);
            const float* lg = logits + r * (K + 1);
            mx = lg[0];
            for (int c = 1; c <= K; ++c) mx = fmaxf(mx, lg[c]);
            sum = 0.f;
            for (int c = 0; c <= K; ++c) sum += expf(lg[c] - mx);
            bool ok = true;
            for (int c = 0; c <= K; ++c) ok = ok && osr_finite(expf(lg[c] - mx) / sum);
            // every class's decoded box must be finite, or the whole row is dropped (valid_mask over dim 1)
            const float w = pb.z - pb.x, h = pb.w - pb.y, cx = pb.x + 0.5f * w, cy = pb.y + 0.5f * h;
            const float* dl = deltas + r * (long long)kbox * 4;
            for (int c = 0; c < kbox && ok; ++c) {
                const float dx = dl[c * 4] / rw.x, dy = dl[c * 4 + 1] / rw.y, dw = fminf(dl[c * 4 + 2] / rw.z, kClamp), dh = fminf(dl[c * 4 + 3] / rw.w, kClamp);
                const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
                ok = osr_finite(pcx - 0.5f * pw) && osr_finite(pcy - 0.5f * ph) && osr_finite(pcx + 0.5f * pw) && osr_finite(pcy + 0.5f * ph) &&
                     dl[c * 4 + 2] == dl[c * 4 + 2] && dl[c * 4 + 3] == dl[c * 4 + 3];  // (fminf drops a NaN delta; torch.clamp keeps it)
            }
            if (ok)
                for (int c = 0; c < K; ++c) nk += (expf(lg[c] - mx) / sum) > score_thresh;
        }
        int tot;
        const int pos = run + osr_block_excl_scan(nk, s_scan, &tot);
        if (nk) {
            const float* lg = logits + r * (K + 1);
            const float* dl = deltas + r * (long long)kbox * 4;
            const float w = pb.z - pb.x, h = pb.w - pb.y, cx = pb.x + 0.5f * w, cy = pb.y + 0.5f * h;
            int o = pos;
            for (int c = 0; c < K; ++c) {
                const float p = expf(lg[c] - mx) / sum;
                if (p > score_thresh) {
                    const int cb = kbox > 1 ? c : 0;
                    const float dx = dl[cb * 4] / rw.x, dy = dl[cb * 4 + 1] / rw.y, dw = fminf(dl[cb * 4 + 2] / rw.z, kClamp), dh = fminf(dl[cb * 4 + 3] / rw.w, kClamp);
                    const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
                    float4 b = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
                    b.x = fminf(fmaxf(b.x, 0.f), iw); b.y = fminf(fmaxf(b.y, 0.f), ih);
                    b.z = fminf(fmaxf(b.z, 0.f), iw); b.w = fminf(fmaxf(b.w, 0.f), ih);
                    const long long q = (long long)img * ccap + o;
                    *reinterpret_cast<float4*>(c_boxes + q * 4) = b;
                    c_scores[q] = p; c_cls[q] = c; c_row[q] = j;
                    ++o;
                }
            }
        }
        run += tot;
    }
    if (tid == 0) c_count[img] = run;
}

extern "C" osr_status osr_fastrcnn_candidates(const float* logits, const float* deltas, int32_t num_classes, int32_t num_bbox_reg_classes,
                                              const float* prop_boxes, const int32_t* prop_count, int32_t n, int32_t seg_rows,
                                              const int32_t* image_hw, const float reg_weights[4], float score_thresh, float* c_boxes,
                                              float* c_scores, int32_t* c_cls, int32_t* c_row, int32_t* c_count, void* stream) {
    OSR_REQUIRE(logits && deltas && prop_boxes && prop_count && image_hw && reg_weights && c_boxes && c_scores && c_cls && c_row && c_count,
                OSR_ERR_INVALID_ARG, "osr_fastrcnn_candidates: null pointer");
    OSR_REQUIRE(n >= 1 && seg_rows >= 1 && num_classes >= 1 && num_classes <= FR_MAX_CLASSES && (num_bbox_reg_classes == 1 || num_bbox_reg_classes == num_classes),
                OSR_ERR_INVALID_ARG, "osr_fastrcnn_candidates: bad sizes (1..%d classes; box regression class-agnostic or per class)", FR_MAX_CLASSES);
    OSR_REQUIRE(reg_weights[0] > 0.f && reg_weights[1] > 0.f && reg_weights[2] > 0.f && reg_weights[3] > 0.f, OSR_ERR_INVALID_ARG,
                "osr_fastrcnn_candidates: regression weights must be positive");
    OSR_REQUIRE((((uintptr_t)prop_boxes | (uintptr_t)c_boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_fastrcnn_candidates: box arrays must be 16-byte aligned");
    hipLaunchKernelGGL(fastrcnn_cand_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, logits, deltas, num_classes, num_bbox_reg_classes, prop_boxes,
                       prop_count, seg_rows, image_hw, make_float4(reg_weights[0], reg_weights[1], reg_weights[2], reg_weights[3]), score_thresh, c_boxes,
                       c_scores, c_cls, c_row, c_count);
    OSR_CHECK_LAUNCH("osr_fastrcnn_candidates");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// final assembly: [unknown..., known...]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ k_boxes, const float* __restrict__ k_scores,
                                                       const int* __restrict__ k_cls, const int* __restrict__ k_keep,
                                                       const int* __restrict__ k_keep_count, long long k_stride, int k_topk,
                                                       const float* __restrict__ u_boxes, const float* __restrict__ u_scores,
                                                       const int* __restrict__ u_keep, const int* __restrict__ u_keep_count,
                                                       long long u_stride, int u_topk, long long unknown_id,
                                                       const long long* __restrict__ class_map, float* __restrict__ out_boxes,
                                                       float* __restrict__ out_scores, long long* __restrict__ out_classes,
                                                       int* __restrict__ out_count) {
    const int img = blockIdx.x, cap = u_topk + k_topk;
    int nu = u_keep_count[img], nk = k_keep_count[img];
    nu = nu < u_topk ? nu : u_topk;
    nk = nk < k_topk ? nk : k_topk;
    for (int i = threadIdx.x; i < cap; i += blockDim.x) {
        const long long o = (long long)img * cap + i;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
        long long c = -1;
        if (i < nu) {
            const long long q = img * u_stride + u_keep[(long long)img * u_topk + i];
            b = *reinterpret_cast<const float4*>(u_boxes + q * 4);
            s = u_scores[q];
            c = unknown_id;
        } else if (i < nu + nk) {
            const long long q = img * k_stride + k_keep[(long long)img * k_topk + (i - nu)];
            b = *reinterpret_cast<const float4*>(k_boxes + q * 4);
            s = k_scores[q];
            c = class_map ? class_map[k_cls[q]] : (long long)k_cls[q];
        }
        *reinterpret_cast<float4*>(out_boxes + o * 4) = b;
        out_scores[o] = s;
        out_classes[o] = c;
    }
    if (threadIdx.x == 0) out_count[img] = nu + nk;
}

extern "C" osr_status osr_assemble_detections(const float* k_boxes, const float* k_scores, const int32_t* k_cls, const int32_t* k_keep,
                                              const int32_t* k_keep_count, int64_t k_stride, int32_t k_topk, const float* u_boxes,
                                              const float* u_scores, const int32_t* u_keep, const int32_t* u_keep_count, int64_t u_stride,
                                              int32_t u_topk, int32_t n, int64_t unknown_id, const int64_t* class_map, float* out_boxes,
                                              float* out_scores, int64_t* out_classes, int32_t* out_count, void* stream) {
    OSR_REQUIRE(k_boxes && k_scores && k_cls && k_keep && k_keep_count && u_boxes && u_scores && u_keep && u_keep_count && out_boxes &&
                    out_scores && out_classes && out_count, OSR_ERR_INVALID_ARG, "osr_assemble_detections: null pointer");
    OSR_REQUIRE(n >= 1 && k_topk >= 1 && u_topk >= 1 && k_stride >= 1 && u_stride >= 1, OSR_ERR_INVALID_ARG, "osr_assemble_detections: bad sizes");
    hipLaunchKernelGGL(assemble_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, k_boxes, k_scores, k_cls, k_keep, k_keep_count,
                       (long long)k_stride, k_topk, u_boxes, u_scores, u_keep, u_keep_count, (long long)u_stride, u_topk, (long long)unknown_id,
                       (const long long*)class_map, out_boxes, out_scores, (long long*)out_classes, out_count);
    OSR_CHECK_LAUNCH("osr_assemble_detections");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// [d2] detector_postprocess: rescale to the requested output resolution, clip, drop empty boxes (order kept)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          const long long* __restrict__ classes, const int* __restrict__ count, int cap,
                                                          const float* __restrict__ scale_xy, const int* __restrict__ out_hw, float* __restrict__ o_boxes,
                                                          float* __restrict__ o_scores, long long* __restrict__ o_classes, int* __restrict__ o_count) {
    __shared__ int s_scan[32];
    const int img = blockIdx.x;
    const int n = min(count[img], cap);
    const float sx = scale_xy[img * 2], sy = scale_xy[img * 2 + 1];
    const float W = (float)out_hw[img * 2 + 1], H = (float)out_hw[img * 2];
    int base = 0;
    for (int i0 = 0; i0 < cap; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        bool keep = false;
        if (i < n) {
            b = *reinterpret_cast<const float4*>(boxes + ((long long)img * cap + i) * 4);
            // Boxes.scale then Boxes.clip: x in [0, W], y in [0, H]; nonempty: w > 0 and h > 0
            b.x = fminf(fmaxf(b.x * sx, 0.f), W); b.z = fminf(fmaxf(b.z * sx, 0.f), W);
            b.y = fminf(fmaxf(b.y * sy, 0.f), H); b.w = fminf(fmaxf(b.w * sy, 0.f), H);
            keep = (b.z - b.x) > 0.f && (b.w - b.y) > 0.f;
        }
        int tot;
        const int pos = base + osr_block_excl_scan(keep ? 1 : 0, s_scan, &tot);
        if (keep) {
            const long long o = (long long)img * cap + pos, q = (long long)img * cap + i;
            *reinterpret_cast<float4*>(o_boxes + o * 4) = b;
            o_scores[o] = scores[q];
            o_classes[o] = classes[q];
        }
        base += tot;
        __syncthreads();
    }
    for (int i = base + threadIdx.x; i < cap; i += blockDim.x) {
        const long long o = (long long)img * cap + i;
        *reinterpret_cast<float4*>(o_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        o_scores[o] = 0.f;
        o_classes[o] = -1;
    }
    if (threadIdx.x == 0) o_count[img] = base;
}

extern "C" osr_status osr_detector_postprocess(const float* boxes, const float* scores, const int64_t* classes, const int32_t* count, int32_t n,
                                               int32_t cap, const float* scale_xy, const int32_t* out_hw, float* out_boxes, float* out_scores,
                                               int64_t* out_classes, int32_t* out_count, void* stream) {
    OSR_REQUIRE(boxes && scores && classes && count && scale_xy && out_hw && out_boxes && out_scores && out_classes && out_count, OSR_ERR_INVALID_ARG,
                "osr_detector_postprocess: null pointer");
    OSR_REQUIRE(n >= 1 && cap >= 1, OSR_ERR_INVALID_ARG, "osr_detector_postprocess: bad n / cap");
    OSR_REQUIRE(boxes != out_boxes, OSR_ERR_INVALID_ARG, "osr_detector_postprocess: in-place operation is not supported");
    hipLaunchKernelGGL(postprocess_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, boxes, scores, (const long long*)classes, count, cap, scale_xy, out_hw,
                       out_boxes, out_scores, (long long*)out_classes, out_count);
    OSR_CHECK_LAUNCH("osr_detector_postprocess");
    return OSR_OK;
}
