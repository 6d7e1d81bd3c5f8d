// Training step, backward half: gradients of the six losses and of the small per-row stages (CF-RPN tail, RoIAlign),
// elementwise helpers and the SGD update. The dense layers' gradients are in osr_conv_bwd.hip.
//
// Conventions: every gradient is d(sum of the weighted losses) / d(tensor) multiplied by `loss_scale` (static loss scaling so
// that fp16 gradient tensors do not underflow; the SGD step divides it out). Where the forward takes a max/min of two values
// the gradient goes to the strictly selected one (ties are measure-zero for real data; torch splits them evenly).
// Reductions over rows are two-stage and fixed-order: results are bitwise reproducible.
#include "osr_common.h"
#include "osr_box_loss.h"
#include "osr_pln_dist.h"

struct TbLevels {
    int num_levels, num_anchors;
    int h[OSR_MAX_LEVELS], w[OSR_MAX_LEVELS], stride[OSR_MAX_LEVELS];
    long long pred_off[OSR_MAX_LEVELS];
    int aoff[OSR_MAX_LEVELS + 1];
    int R;
};
static bool tb_fill(const osr_rpn_levels* in, TbLevels* o) {
    if (!in || in->num_levels < 1 || in->num_levels > OSR_MAX_LEVELS || in->num_anchors < 1) return false;
    o->num_levels = in->num_levels; o->num_anchors = in->num_anchors;
    long long a = 0;
    for (int l = 0; l < in->num_levels; ++l) {
        if (in->h[l] < 1 || in->w[l] < 1 || in->stride[l] < 1) return false;
        o->h[l] = in->h[l]; o->w[l] = in->w[l]; o->stride[l] = in->stride[l]; o->pred_off[l] = in->offset[l];
        o->aoff[l] = (int)a;
        a += (long long)in->h[l] * in->w[l] * in->num_anchors;
        if (a > (1ll << 30)) return false;
    }
    o->aoff[in->num_levels] = (int)a;
    o->R = (int)a;
    return true;
}
__device__ __forceinline__ float4 tb_anchor(const TbLevels& lv, const float* __restrict__ cell, int r, int* level, int* cell_idx) {
    int l = 0;
    while (l + 1 < lv.num_levels && r >= lv.aoff[l + 1]) ++l;
    const int idx = r - lv.aoff[l], A = lv.num_anchors, a = idx % A, c = idx / A;
    const float sx = (float)(c % lv.w[l]) * (float)lv.stride[l], sy = (float)(c / lv.w[l]) * (float)lv.stride[l];
    const float* ca = cell + ((long long)l * A + a) * 4;
    *level = l; *cell_idx = idx;
    return make_float4(sx + ca[0], sy + ca[1], sx + ca[2], sy + ca[3]);
}

// ------------------------------------------------------------------------------------------------------
// ClsFreeRPN.losses backward: gradient w.r.t. the head's five pre-activation outputs per anchor
// (ltrb deltas, centerness logit). out5 is level-major like the predictions; every row is written.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rpn_losses_bwd_kernel(TbLevels lv, const float* __restrict__ cell, int n, const float* __restrict__ pred_deltas,
                                                             const float* __restrict__ pred_ctr, const signed char* __restrict__ labels_reg,
                                                             const signed char* __restrict__ labels_obj, const float* __restrict__ matched_boxes,
                                                             const float* __restrict__ ctr_target, float s_loc, float s_ctr, int box_type, float box_beta,
                                                             float ctr_beta, float* __restrict__ out5) {
    const long long total = (long long)n * lv.R;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(i / lv.R), r = (int)(i - (long long)img * lv.R);
        int l, ci;
        const float4 a = tb_anchor(lv, cell, r, &l, &ci);
        const long long pi = lv.pred_off[l] + (long long)img * (lv.aoff[l + 1] - lv.aoff[l]) + ci;
        float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const signed char lr = labels_reg[i], lo = labels_obj[i];
        if (lr == 1) {
            const float4 d = *reinterpret_cast<const float4*>(pred_deltas + pi * 4);
            const float cx = 0.5f * (a.x + a.z), cy = 0.5f * (a.y + a.w), aw = a.z - a.x, ah = a.w - a.y;
            const float4 gt = *reinterpret_cast<const float4*>(matched_boxes + i * 4);
            if (box_type == OSR_LOSS_SMOOTH_L1) {
                g[0] = s_loc * osr_smooth_l1_grad(d.x - (cx - gt.x) / aw, box_beta);
                g[1] = s_loc * osr_smooth_l1_grad(d.y - (cy - gt.y) / ah, box_beta);
                g[2] = s_loc * osr_smooth_l1_grad(d.z - (gt.z - cx) / aw, box_beta);
                g[3] = s_loc * osr_smooth_l1_grad(d.w - (gt.w - cy) / ah, box_beta);
            } else {
                const float4 pb = make_float4(cx - fmaxf(d.x, 0.f) * aw, cy - fmaxf(d.y, 0.f) * ah, cx + fmaxf(d.z, 0.f) * aw, cy + fmaxf(d.w, 0.f) * ah);
                float dp[4];
                (void)osr_box_loss<true>(box_type, pb, gt, dp);
                const float db[4] = {d.x > 0.f ? -aw : 0.f, d.y > 0.f ? -ah : 0.f, d.z > 0.f ? aw : 0.f, d.w > 0.f ? ah : 0.f};  // d box / d delta (through the ReLU)
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = s_loc * dp[q] * db[q];
            }
        }
        if (lo != -1) {
            const float c = pred_ctr[pi];
            g[4] = s_ctr * osr_smooth_l1_grad(c - ctr_target[i], ctr_beta) * c * (1.f - c);  // through the sigmoid
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) out5[pi * 5 + q] = g[q];
    }
}

extern "C" osr_status osr_rpn_losses_bwd_ex(const osr_rpn_levels* lvl, const float* cell_anchors, int32_t n, const float* pred_deltas,
                                         const float* pred_ctr, const int8_t* labels_reg, const int8_t* labels_obj, const float* matched_boxes,
                                         const float* ctr_target, float loc_weight, float ctr_weight, int32_t batch_size_per_image, float loss_scale,
                                            const osr_loss_options* opt, float* d_out5, void* stream) {
    const int box_type = opt ? opt->box_loss_type : OSR_LOSS_IOU;
    const float box_beta = opt ? opt->box_smooth_l1_beta : 0.f, ctr_beta = opt ? opt->aux_smooth_l1_beta : 0.f;
    OSR_REQUIRE(box_type >= OSR_LOSS_IOU && box_type <= OSR_LOSS_CIOU && box_beta >= 0.f && ctr_beta >= 0.f, OSR_ERR_INVALID_ARG,
                "osr_rpn_losses_bwd: bad loss options (type %d)", box_type);
    TbLevels lv;
    OSR_REQUIRE(tb_fill(lvl, &lv), OSR_ERR_INVALID_ARG, "osr_rpn_losses_bwd: bad level table");
    OSR_REQUIRE(cell_anchors && pred_deltas && pred_ctr && labels_reg && labels_obj && matched_boxes && ctr_target && d_out5, OSR_ERR_INVALID_ARG,
                "osr_rpn_losses_bwd: null pointer");
    OSR_REQUIRE(n >= 1 && batch_size_per_image >= 1, OSR_ERR_INVALID_ARG, "osr_rpn_losses_bwd: bad n / batch size");
    OSR_REQUIRE((((uintptr_t)pred_deltas | (uintptr_t)matched_boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_rpn_losses_bwd: box arrays must be 16-byte aligned");
    const float norm = (float)batch_size_per_image * (float)n;
    hipLaunchKernelGGL(rpn_losses_bwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, lv, cell_anchors, n, pred_deltas, pred_ctr,
                       (const signed char*)labels_reg, (const signed char*)labels_obj, matched_boxes, ctr_target, loss_scale * loc_weight / norm,
                       loss_scale * ctr_weight / norm, box_type, box_beta, ctr_beta, d_out5);
    OSR_CHECK_LAUNCH("osr_rpn_losses_bwd");
    return OSR_OK;
}

extern "C" osr_status osr_rpn_losses_bwd(const osr_rpn_levels* lvl, const float* cell_anchors, int32_t n, const float* pred_deltas,
                                         const float* pred_ctr, const int8_t* labels_reg, const int8_t* labels_obj, const float* matched_boxes,
                                         const float* ctr_target, float loc_weight, float ctr_weight, int32_t batch_size_per_image, float loss_scale,
                                         float* d_out5, void* stream) {
    return osr_rpn_losses_bwd_ex(lvl, cell_anchors, n, pred_deltas, pred_ctr, labels_reg, labels_obj, matched_boxes, ctr_target, loc_weight, ctr_weight,
                                 batch_size_per_image, loss_scale, nullptr, d_out5, stream);
}

// ------------------------------------------------------------------------------------------------------
// ClsFreeRPNHead tail backward (classification_free_rpn.py:159-161): t -> u = t / max(||t||, eps) -> o_q = w_q . u + b_q.
// Given d o (T,5): dt (masked by t > 0, i.e. through the ReLU of the 3x3 conv), dW (5,256), db (5).
// One wave per row, 4 channels per lane; rows whose five gradients are all zero (all but the sampled anchors) only write zeros.
// ------------------------------------------------------------------------------------------------------
#define TAILB_BLOCKS 512
template <class TI>
__global__ __launch_bounds__(256) void cfrpn_tail_bwd_kernel(const TI* __restrict__ t, long long T, const float* __restrict__ w_tail,
                                                             const float* __restrict__ d_out5, TI* __restrict__ dt, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wid, nw = gridDim.x * 4;
    float w[5][4];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[q][e] = w_tail[q * 256 + lane * 4 + e];
    float dw[5][4], db[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        db[q] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[q][e] = 0.f;
    }
    for (long long r = gw; r < T; r += nw) {
        float go[5];
        bool any = false;
#pragma unroll
        for (int q = 0; q < 5; ++q) { go[q] = d_out5[r * 5 + q]; any |= go[q] != 0.f; }
        TI* drow = dt + r * 256 + lane * 4;
        if (!any) {  // uniform across the wave: the row's gradients are the same for every lane
#pragma unroll
            for (int e = 0; e < 4; ++e) drow[e] = osr_from_float<TI>(0.f);
            continue;
        }
        float tv[4], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { tv[e] = osr_to_float(t[r * 256 + lane * 4 + e]); ss += tv[e] * tv[e]; }
        ss = osr_wave_sum(ss);
        const float nrm = sqrtf(ss);
        const bool clamped = nrm <= 1e-12f;
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);
        float u[4], du[4], dot = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u[e] = tv[e] * inv;
            du[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 5; ++q) du[e] += go[q] * w[q][e];
            dot += u[e] * du[e];
#pragma unroll
            for (int q = 0; q < 5; ++q) dw[q][e] += go[q] * u[e];
        }
        dot = clamped ? 0.f : osr_wave_sum(dot);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) db[q] += go[q];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g = (du[e] - u[e] * dot) * inv;
            drow[e] = osr_from_float<TI>(tv[e] > 0.f ? g : 0.f);
        }
    }
    // partial[gw][5][256] + [gw][5]
    float* pw = partial + (long long)gw * (5 * 256 + 8);
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[q * 256 + lane * 4 + e] = dw[q][e];
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) pw[5 * 256 + q] = db[q];
    }
}

// 64 columns per workgroup, 16 groups of partials per column summed side by side (each a contiguous run, eight loads in flight), then the
// 16 group sums in a fixed order: deterministic, and 2048 partials deep instead of one 2048-long dependent chain per thread (0.61 ms)
#define TAILR_GROUPS 16
__global__ __launch_bounds__(64 * TAILR_GROUPS) void cfrpn_tail_bwd_reduce(const float* __restrict__ partial, int nparts, int accumulate, float* __restrict__ dw,
                                                                           float* __restrict__ db) {
    __shared__ float s_part[TAILR_GROUPS][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;  // 0 .. 5*256+5
    const int per = (nparts + TAILR_GROUPS - 1) / TAILR_GROUPS, p0 = g * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    float s = 0.f;
    if (i < 5 * 256 + 5) {
#pragma unroll 8
        for (int p = p0; p < p1; ++p) s += partial[(long long)p * (5 * 256 + 8) + i];
    }
    s_part[g][col] = s;
    __syncthreads();
    if (g != 0 || i >= 5 * 256 + 5) return;
    s = s_part[0][col];
#pragma unroll
    for (int k = 1; k < TAILR_GROUPS; ++k) s += s_part[k][col];
    if (i < 5 * 256) dw[i] = (accumulate ? dw[i] : 0.f) + s;
    else db[i - 5 * 256] = (accumulate ? db[i - 5 * 256] : 0.f) + s;
}

extern "C" int64_t osr_cfrpn_tail_bwd_workspace_bytes(void) { return (int64_t)TAILB_BLOCKS * 4 * (5 * 256 + 8) * 4; }

extern "C" osr_status osr_cfrpn_tail_bwd(const void* t, int32_t dtype, int64_t rows, const float* w_tail, const float* d_out5, void* dt, float* dw_tail,
                                         float* db_tail, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(t && w_tail && d_out5 && dt && dw_tail && db_tail && workspace, OSR_ERR_INVALID_ARG, "osr_cfrpn_tail_bwd: null pointer");
    OSR_REQUIRE(rows >= 1 && (dtype == OSR_F16 || dtype == OSR_BF16), OSR_ERR_INVALID_ARG, "osr_cfrpn_tail_bwd: bad rows / dtype");
    OSR_REQUIRE(workspace_bytes >= osr_cfrpn_tail_bwd_workspace_bytes(), OSR_ERR_WORKSPACE, "osr_cfrpn_tail_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OSR_F16)
        hipLaunchKernelGGL(cfrpn_tail_bwd_kernel<f16_t>, dim3(TAILB_BLOCKS), dim3(256), 0, st, (const f16_t*)t, (long long)rows, w_tail, d_out5, (f16_t*)dt, (float*)workspace);
    else
        hipLaunchKernelGGL(cfrpn_tail_bwd_kernel<bf16_t>, dim3(TAILB_BLOCKS), dim3(256), 0, st, (const bf16_t*)t, (long long)rows, w_tail, d_out5, (bf16_t*)dt, (float*)workspace);
    OSR_CHECK_LAUNCH("osr_cfrpn_tail_bwd");
    hipLaunchKernelGGL(cfrpn_tail_bwd_reduce, dim3((5 * 256 + 5 + 63) / 64), dim3(64 * TAILR_GROUPS), 0, st, (const float*)workspace, TAILB_BLOCKS * 4, accumulate, dw_tail, db_tail);
    OSR_CHECK_LAUNCH("osr_cfrpn_tail_bwd(reduce)");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// RoI-head losses backward
// ------------------------------------------------------------------------------------------------------
// rows counted = rows with class >= 0 (as in the forward); one tiny kernel counts them so that both passes agree
__global__ void count_rows_kernel(const long long* __restrict__ cls, long long m, int lo, int hi_excl, float* __restrict__ out) {
    __shared__ int s_cnt[256];
    int c = 0;
    for (long long i = threadIdx.x; i < m; i += blockDim.x) c += (cls[i] >= lo && (hi_excl < 0 || cls[i] < hi_excl)) ? 1 : 0;
    s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)s_cnt[0];
}

__global__ __launch_bounds__(256) void roi_box_losses_bwd_kernel(const float* __restrict__ pred, int pstride, const float* __restrict__ prop,
                                                                 const float* __restrict__ gtb, const long long* __restrict__ cls,
                                                                 const float* __restrict__ gt_iou, long long m, int num_classes, float wx, float wy, float ww,
                                                                 float wh, float s_box, float s_iou, int box_type, float box_beta, float iou_beta,
                                                                 const float* __restrict__ rows, float* __restrict__ d_pred) {
    const float r = fmaxf(rows[0], 1.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const long long c = cls[i];
        if (c >= 0 && c < num_classes) {
            const float4 s = *reinterpret_cast<const float4*>(prop + i * 4), t = *reinterpret_cast<const float4*>(gtb + i * 4);
            const float* d = pred + i * pstride;
            const float sw = s.z - s.x, sh = s.w - s.y, scx = s.x + 0.5f * sw, scy = s.y + 0.5f * sh;
            const float tw = t.z - t.x, th = t.w - t.y, tcx = t.x + 0.5f * tw, tcy = t.y + 0.5f * th;
            if (box_type == OSR_LOSS_SMOOTH_L1) {
                const float tg[4] = {wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * logf(tw / sw), wh * logf(th / sh)};
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = osr_smooth_l1_grad(d[q] - tg[q], box_beta) * s_box / r;
            } else {  // through [d2] Box2BoxTransform.apply_deltas: centre = d/w * size + centre, size = exp(min(d/w, clamp)) * size
                const float kClamp = 4.135166556742356f;
                const float ew = d[2] / ww, eh = d[3] / wh;
                const float pw = expf(fminf(ew, kClamp)) * sw, ph = expf(fminf(eh, kClamp)) * sh;
                const float pcx = d[0] / wx * sw + scx, pcy = d[1] / wy * sh + scy;
                float dp[4];
                (void)osr_box_loss<true>(box_type, make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph), t, dp);
                g[0] = (dp[0] + dp[2]) * sw / wx * s_box / r;
                g[1] = (dp[1] + dp[3]) * sh / wy * s_box / r;
                g[2] = ew < kClamp ? 0.5f * (dp[2] - dp[0]) * pw / ww * s_box / r : 0.f;
                g[3] = eh < kClamp ? 0.5f * (dp[3] - dp[1]) * ph / wh * s_box / r : 0.f;
            }
            const float sg = 1.0f / (1.0f + expf(-d[4]));
            g[4] = osr_smooth_l1_grad(sg - gt_iou[i], iou_beta) * sg * (1.f - sg) * s_iou / r;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) d_pred[i * 5 + q] = g[q];
    }
}

extern "C" osr_status osr_roi_box_losses_bwd_ex(const float* pred5, const float* proposal_boxes, const float* gt_boxes, const int64_t* gt_classes,
                                             const float* gt_iou, int64_t m, int32_t num_classes, const float reg_weights[4], float box_weight,
                                             float iou_weight, float loss_scale, const osr_loss_options* opt, float* d_pred5, void* workspace,
                                                int64_t workspace_bytes, void* stream) {
    const int box_type = opt ? opt->box_loss_type : OSR_LOSS_SMOOTH_L1;
    const float box_beta = opt ? opt->box_smooth_l1_beta : 0.f, iou_beta = opt ? opt->aux_smooth_l1_beta : 0.f;
    OSR_REQUIRE(box_type >= OSR_LOSS_IOU && box_type <= OSR_LOSS_CIOU && box_beta >= 0.f && iou_beta >= 0.f, OSR_ERR_INVALID_ARG,
                "osr_roi_box_losses_bwd: bad loss options (type %d)", box_type);
    OSR_REQUIRE(pred5 && proposal_boxes && gt_boxes && gt_classes && gt_iou && reg_weights && d_pred5 && workspace, OSR_ERR_INVALID_ARG,
                "osr_roi_box_losses_bwd: null pointer");
    OSR_REQUIRE(m >= 1 && workspace_bytes >= 16, OSR_ERR_INVALID_ARG, "osr_roi_box_losses_bwd: bad m / workspace (16 bytes)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(count_rows_kernel, dim3(1), dim3(256), 0, st, (const long long*)gt_classes, (long long)m, 0, -1, (float*)workspace);
    OSR_CHECK_LAUNCH("osr_roi_box_losses_bwd(count)");
    hipLaunchKernelGGL(roi_box_losses_bwd_kernel, dim3(256), dim3(256), 0, st, pred5, 5, proposal_boxes, gt_boxes, (const long long*)gt_classes, gt_iou, (long long)m,
                       num_classes, reg_weights[0], reg_weights[1], reg_weights[2], reg_weights[3], loss_scale * box_weight, loss_scale * iou_weight,
                       box_type, box_beta, iou_beta, (const float*)workspace, d_pred5);
    OSR_CHECK_LAUNCH("osr_roi_box_losses_bwd");
    return OSR_OK;
}

extern "C" osr_status osr_roi_box_losses_bwd(const float* pred5, const float* proposal_boxes, const float* gt_boxes, const int64_t* gt_classes,
                                             const float* gt_iou, int64_t m, int32_t num_classes, const float reg_weights[4], float box_weight,
                                             float iou_weight, float loss_scale, float* d_pred5, void* workspace, int64_t workspace_bytes, void* stream) {
    return osr_roi_box_losses_bwd_ex(pred5, proposal_boxes, gt_boxes, gt_classes, gt_iou, m, num_classes, reg_weights, box_weight, iou_weight, loss_scale,
                                     nullptr, d_pred5, workspace, workspace_bytes, stream);
}

// softmax cross entropy: d logits = (softmax - onehot) * weight / count over the rows with a valid target
__global__ __launch_bounds__(256) void ce_count_kernel(const long long* __restrict__ cls, long long m, int num_classes, int K, float* __restrict__ out) {
    __shared__ int s_cnt[256];
    int c = 0;
    for (long long i = threadIdx.x; i < m; i += blockDim.x) {
        const long long v = cls[i];
        c += ((v >= 0 && v < K) || v == num_classes) ? 1 : 0;
    }
    s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)s_cnt[0];
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long long m, int nc, const long long* __restrict__ cls, int num_classes,
                                                     int K, float s, const float* __restrict__ count, float* __restrict__ d_logits) {
    const float cnt = count[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const long long c = cls[i];
        const int t = (c >= 0 && c < K) ? (int)c : (c == num_classes ? K : -1);
        float* dl = d_logits + i * nc;
        if (t < 0 || cnt <= 0.f) {
            for (int j = 0; j < nc; ++j) dl[j] = 0.f;
            continue;
        }
        const float* lg = logits + i * nc;
        float mx = lg[0];
        for (int j = 1; j < nc; ++j) mx = fmaxf(mx, lg[j]);
        float sum = 0.f;
        for (int j = 0; j < nc; ++j) sum += expf(lg[j] - mx);
        for (int j = 0; j < nc; ++j) dl[j] = (expf(lg[j] - mx) / sum - (j == t ? 1.f : 0.f)) * s / cnt;
    }
}

extern "C" osr_status osr_softmax_ce_loss_bwd(const float* logits, int64_t m, int32_t num_known, const int64_t* gt_classes, int32_t num_classes,
                                              float loss_weight, float loss_scale, float* d_logits, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(logits && gt_classes && d_logits && workspace, OSR_ERR_INVALID_ARG, "osr_softmax_ce_loss_bwd: null pointer");
    OSR_REQUIRE(m >= 1 && num_known >= 1 && workspace_bytes >= 16, OSR_ERR_INVALID_ARG, "osr_softmax_ce_loss_bwd: bad sizes / workspace (16 bytes)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, st, (const long long*)gt_classes, (long long)m, num_classes, num_known, (float*)workspace);
    OSR_CHECK_LAUNCH("osr_softmax_ce_loss_bwd(count)");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(256), dim3(256), 0, st, logits, (long long)m, num_known + 1, (const long long*)gt_classes, num_classes, num_known,
                       loss_scale * loss_weight, (const float*)workspace, d_logits);
    OSR_CHECK_LAUNCH("osr_softmax_ce_loss_bwd");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// PLN hinge loss backward: d emb (m,d) and d prototypes (K,d) w.r.t. the RAW (un-normalised) prototypes
// ------------------------------------------------------------------------------------------------------
// pass 1 (wave per row): coefficients of the two prototypes a foreground row pulls on, and the row's d emb
template <int NJ>
__global__ __launch_bounds__(256) void pln_bwd_rows_kernel(const float* __restrict__ emb, long long m, int d, const float* __restrict__ protos, int K, int R,
                                                           int dist_type, const long long* __restrict__ cls, const float* __restrict__ ious, float iou_thr,
                                                           float alpha, float beta, float s, const float* __restrict__ rows, float* __restrict__ d_emb,
                                                           int* __restrict__ pair_idx, float* __restrict__ pair_coef, float* __restrict__ pair_dist,
                                                           float* __restrict__ row_inv) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];  // normalised prototypes [K * R][d]
    const int KR = K * R;
    for (int k = threadIdx.x >> 6; k < KR; k += blockDim.x >> 6) {
        const int lane = threadIdx.x & 63;
        float ss = 0.f;
        for (int i = lane; i < d; i += 64) { const float x = protos[k * d + i]; ss += x * x; }
        ss = osr_wave_sum(ss);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        for (int i = lane; i < d; i += 64) s_p[k * d + i] = protos[k * d + i] * inv;
    }
    __syncthreads();
    const float sc = s / fmaxf(rows[0], 1.0f);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int W = (int)gridDim.x * nw, w = (int)blockIdx.x * nw + wid;
    for (long long k = 0; k * W < m; ++k) {
        const long long r = osr_pln_row(k, w, W);
        if (r >= m) continue;
        const long long y = cls[r];
        const float iou = ious[r];  // (both loads up front: one round trip)
        float* de = d_emb + r * d;
        const bool fg = y >= 0 && y < K && iou > iou_thr;
        int i0 = -1, i1 = -1;
        float c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f, rinv = 0.f;
        if (fg) {
            const float* e = emb + r * d;
            const bool in_regs = d <= NJ * 64;  // (wave-uniform) the normalised row lives in registers for the class loop
            float ehr[NJ];  // the row, read once: ||e|| from these registers, then scaled in place
            float ss = 0.f;
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int i = lane + 64 * j; ehr[j] = i < d ? e[i] : 0.f; ss += ehr[j] * ehr[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) ehr[j] = 0.f;
                for (int i = lane; i < d; i += 64) { const float x = e[i]; ss += x * x; }
            }
            ss = osr_wave_sum(ss);
            const float nrm = sqrtf(ss), inv = 1.0f / fmaxf(nrm, 1e-12f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) ehr[j] = ehr[j] * inv;
            // own class: its nearest prototype (i0, intra); other classes: the nearest prototype of the nearest class (i1, inter)
            float intra = 0.f, inter = 1000.f;
            int pstar = -1, ystar = -1;
            if (in_regs) {  // the prototypes four at a time, in order (osr_pln_distance_reg4); same comparisons in the same order as the class loop below
                float dist = 0.f;
                int arg = 0;
                for (int k0 = 0; k0 < KR; k0 += 4) {
                    float dq[4];
                    osr_pln_distance_reg4<NJ>(ehr, s_p + (size_t)k0 * d, d, KR - k0 < 4 ? KR - k0 : 4, lane, dist_type, dq);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kr = k0 + t;
                        if (kr < KR) {
                            const int c = kr / R, q = kr - c * R;
                            if (q == 0 || dq[t] < dist) { dist = dq[t]; arg = kr; }
                            if (q == R - 1) {
                                if (c == (int)y) { intra = dist; ystar = arg; }
                                else if (dist < inter) { inter = dist; pstar = arg; }
                            }
                        }
                    }
                }
            } else {
                for (int c = 0; c < K; ++c) {
                    float dist = 0.f;
                    int arg = c * R;
                    for (int q = 0; q < R; ++q) {
                        const float dq = osr_pln_distance([&](int i) { return e[i] * inv; }, s_p + (size_t)(c * R + q) * d, d, lane, dist_type);
                        if (q == 0 || dq < dist) { dist = dq; arg = c * R + q; }
                    }
                    if (c == (int)y) { intra = dist; ystar = arg; }
                    else if (dist < inter) { inter = dist; pstar = arg; }
                }
            }
            // dL/dD_y = sc [D_y > alpha];  dL/dD_c* = -sc [beta > D_c*]
            const float gy = intra - alpha > 0.f ? sc : 0.f, gc = (pstar >= 0 && beta - inter > 0.f) ? -sc : 0.f;
            // d ehat = gy dD/dehat(p_y*) + gc dD/dehat(p_c*);  d e = (d ehat - ehat (ehat . d ehat)) / ||e||
            float dot = 0.f;
            if (in_regs) {
                float dhr[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int i = lane + 64 * j;
                    dhr[j] = 0.f;
                    if (i < d) {
                        dhr[j] = gy * osr_pln_ddist_da(ehr[j], s_p[ystar * d + i], intra, dist_type) +
                                 (pstar >= 0 ? gc * osr_pln_ddist_da(ehr[j], s_p[pstar * d + i], inter, dist_type) : 0.f);
                        dot += ehr[j] * dhr[j];
                    }
                }
                dot = osr_wave_sum(dot);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int i = lane + 64 * j;
                    if (i < d) de[i] = nrm > 1e-12f ? (dhr[j] - ehr[j] * dot) * inv : dhr[j] * inv;
                }
            } else {
                for (int i = lane; i < d; i += 64) {
                    const float eh = e[i] * inv;
                    const float dh = gy * osr_pln_ddist_da(eh, s_p[ystar * d + i], intra, dist_type) +
                                     (pstar >= 0 ? gc * osr_pln_ddist_da(eh, s_p[pstar * d + i], inter, dist_type) : 0.f);
                    dot += eh * dh;
                }
                dot = osr_wave_sum(dot);
                for (int i = lane; i < d; i += 64) {
                    const float eh = e[i] * inv;
                    const float dh = gy * osr_pln_ddist_da(eh, s_p[ystar * d + i], intra, dist_type) +
                                     (pstar >= 0 ? gc * osr_pln_ddist_da(eh, s_p[pstar * d + i], inter, dist_type) : 0.f);
                    de[i] = nrm > 1e-12f ? (dh - eh * dot) * inv : dh * inv;
                }
            }
            // d phat_y* += gy dD/dphat ; d phat_c* += gc dD/dphat (pass 2)
            i0 = ystar; c0 = gy; d0 = intra; i1 = pstar; c1 = gc; d1 = inter; rinv = inv;
        } else {
            for (int i = lane; i < d; i += 64) de[i] = 0.f;
        }
        if (lane == 0) {
            pair_idx[r * 2] = i0; pair_idx[r * 2 + 1] = i1;
            pair_coef[r * 2] = c0; pair_coef[r * 2 + 1] = c1;
            pair_dist[r * 2] = d0; pair_dist[r * 2 + 1] = d1;
            row_inv[r] = rinv;
        }
    }
}

// pass 2 (one workgroup per prototype): d phat_k = sum over rows (fixed order) coef * dD/dphat_k(ehat_row) + the centre term, then
// the projection through the normalisation of the raw prototype
__global__ __launch_bounds__(1024) void pln_bwd_protos_kernel(const float* __restrict__ emb, long long m, int d, const float* __restrict__ protos, int K, int R,
                                                             int dist_type, const int* __restrict__ pair_idx, const float* __restrict__ pair_coef,
                                                             const float* __restrict__ pair_dist, const float* __restrict__ row_inv, float alpha, float beta,
                                                             float s, const float* __restrict__ rows, int accumulate, float* __restrict__ d_protos) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];  // normalised prototypes [KR][d], then KR norms, KR argmins, KR centre distances, d staged
    const int KR = K * R;
    float* s_nrm = s_p + KR * d;
    int* s_arg = reinterpret_cast<int*>(s_nrm + KR);
    float* s_cd = reinterpret_cast<float*>(s_arg + KR);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = wid; k < KR; k += nw) {
        float ss = 0.f;
        for (int i = lane; i < d; i += 64) { const float x = protos[k * d + i]; ss += x * x; }
        ss = osr_wave_sum(ss);
        const float nrm = sqrtf(ss), inv = 1.0f / fmaxf(nrm, 1e-12f);
        for (int i = lane; i < d; i += 64) s_p[k * d + i] = protos[k * d + i] * inv;
        if (lane == 0) s_nrm[k] = nrm;
    }
    __syncthreads();
    for (int k = wid; k < KR; k += nw) {  // centre term: nearest prototype of another class, for every prototype
        float cd = 1000.f;
        int arg = -1;
        for (int j = 0; j < KR; ++j) {
            if (j / R == k / R) continue;
            const float* pk = s_p + (size_t)k * d;
            const float dj = osr_pln_distance([&](int i) { return pk[i]; }, s_p + (size_t)j * d, d, lane, dist_type);
            if (dj < cd) { cd = dj; arg = j; }
        }
        if (lane == 0) { s_arg[k] = arg; s_cd[k] = cd; }
    }
    __syncthreads();
    const int k = blockIdx.x;
    const float sc = s / fmaxf(rows[0], 1.0f);
    // Rows whose (intra, inter) prototype pair names k, in row order, PLN_LIST rows at a time: a ballot per wave and the waves' counts
    // give every hit its place in the list, and the thread that found the row parks its scalars (coefficients, distances, 1 / ||e||)
    // beside it, so the channel loop's only global load is the embedding itself. The loop visits only those rows (a few hundred of
    // the 8192): the thread groups (256 channels x blockDim / 256 groups) take every ng-th list entry each with several loads in
    // flight -- one workgroup per prototype used to walk the list as one dependent chain, 0.25 ms -- and the groups' sums are added
    // in group order at the end: a fixed order, so the result is reproducible.
    constexpr int PLN_LIST = 512;
    __shared__ int s_list[PLN_LIST], s_flag[PLN_LIST];
    __shared__ float s_c0[PLN_LIST], s_c1[PLN_LIST], s_d0[PLN_LIST], s_d1[PLN_LIST], s_ri[PLN_LIST];
    __shared__ int s_wcnt[16];
    const int ct = threadIdx.x & 255, grp = threadIdx.x >> 8, ng = (int)blockDim.x >> 8;
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};  // this thread owns channels ct, ct + 256, ... (d <= 1024) of its group's share of the rows
    for (long long base = 0; base < m; base += PLN_LIST) {
        const int chunk = (int)((m - base) < PLN_LIST ? (m - base) : PLN_LIST);
        const int rr = (int)threadIdx.x;
        int i0 = -1, i1 = -1;
        if (rr < chunk) { i0 = pair_idx[(base + rr) * 2]; i1 = pair_idx[(base + rr) * 2 + 1]; }
        const bool hit = i0 == k || i1 == k;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_wcnt[wid] = __popcll(bal);
        __syncthreads();
        int off = 0, nlist = 0;
        for (int w = 0; w < nw; ++w) { const int c = s_wcnt[w]; off += w < wid ? c : 0; nlist += c; }
        if (hit) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            const long long r = base + rr;
            s_list[pos] = rr;
            s_flag[pos] = (i0 == k ? 1 : 0) | (i1 == k ? 2 : 0);
            s_c0[pos] = pair_coef[r * 2]; s_c1[pos] = pair_coef[r * 2 + 1];
            s_d0[pos] = pair_dist[r * 2]; s_d1[pos] = pair_dist[r * 2 + 1];
            s_ri[pos] = row_inv[r];
        }
        __syncthreads();
#pragma unroll 4
        for (int li = grp; li < nlist; li += ng) {
            const long long r = base + s_list[li];
            const int f = s_flag[li];
            const float ri = s_ri[li], c0 = s_c0[li], c1 = s_c1[li], d0 = s_d0[li], d1 = s_d1[li];
            int q = 0;
            for (int ch = ct; ch < d; ch += 256, ++q) {
                const float pk = s_p[k * d + ch];
                const float eh = emb[r * d + ch] * ri;  // ehat of the row
                if (f & 1) acc4[q] += c0 * osr_pln_ddist_db(eh, pk, d0, dist_type);
                if (f & 2) acc4[q] += c1 * osr_pln_ddist_db(eh, pk, d1, dist_type);
            }
        }
        __syncthreads();
    }
    float* dph = s_p + KR * d + 3 * KR;  // d phat_k staged behind the tables
    for (int gi = 0; gi < ng; ++gi) {  // the groups' sums, in group order
        if (grp == gi) {
            int q = 0;
            for (int ch = ct; ch < d; ch += 256, ++q) dph[ch] = (gi == 0 ? 0.f : dph[ch]) + acc4[q];
        }
        __syncthreads();
    }
    for (int ch = threadIdx.x; ch < d; ch += blockDim.x) {
        float acc = dph[ch];
        // centre term: L += sc * relu(alpha + beta - cd_j) for every prototype j; cd_j = dist(phat_j, phat_arg(j))
        //   dL/dcd_j = -sc [alpha + beta > cd_j];  d phat_k gets dcd_j/dphat_j when k == j and dcd_j/dphat_arg(j) when k == arg(j)
        for (int j = 0; j < KR; ++j) {
            if (!(alpha + beta - s_cd[j] > 0.f) || s_arg[j] < 0) continue;
            const float pj = s_p[j * d + ch], pa = s_p[s_arg[j] * d + ch];
            if (j == k) acc += -sc * osr_pln_ddist_da(pj, pa, s_cd[j], dist_type);
            if (s_arg[j] == k) acc += -sc * osr_pln_ddist_db(pj, pa, s_cd[j], dist_type);
        }
        dph[ch] = acc;
    }
    __syncthreads();
    // projection: d p = (d phat - phat (phat . d phat)) / ||p||
    __shared__ float s_dot[16];
    float part = 0.f;
    for (int ch = threadIdx.x; ch < d; ch += blockDim.x) part += s_p[k * d + ch] * dph[ch];
    part = osr_wave_sum(part);
    if (lane == 0) s_dot[wid] = part;
    __syncthreads();
    float dot = 0.f;
    for (int w = 0; w < nw; ++w) dot += s_dot[w];
    const float nrm = s_nrm[k], inv = 1.0f / fmaxf(nrm, 1e-12f);
    for (int ch = threadIdx.x; ch < d; ch += blockDim.x) {
        const float g = nrm > 1e-12f ? (dph[ch] - s_p[k * d + ch] * dot) * inv : dph[ch] * inv;
        d_protos[k * d + ch] = (accumulate ? d_protos[k * d + ch] : 0.f) + g;
    }
}

extern "C" int64_t osr_pln_loss_bwd_workspace_bytes(int64_t m) { return 16 + m * 28; }

extern "C" osr_status osr_pln_loss_bwd_ex(const float* emb, int64_t m, int32_t d, const float* protos_raw, int32_t num_known, int32_t reps,
                                          int32_t distance_type, const int64_t* gt_classes, const float* ious, float iou_thr, float alpha, float beta,
                                          float loss_weight, float loss_scale, float* d_emb, float* d_protos, int32_t accumulate_protos, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(emb && protos_raw && gt_classes && ious && d_emb && d_protos && workspace, OSR_ERR_INVALID_ARG, "osr_pln_loss_bwd: null pointer");
    const long long kr = (long long)num_known * reps;
    OSR_REQUIRE(m >= 1 && d >= 1 && d <= 1024 && num_known >= 2 && reps >= 1 && (kr * d + 3 * kr + d) * 4 <= 144 * 1024, OSR_ERR_UNSUPPORTED,
                "osr_pln_loss_bwd: bad sizes (the prototypes and their tables must fit 144 KB of LDS)");
    OSR_REQUIRE(distance_type >= OSR_DIST_COS && distance_type <= OSR_DIST_L2, OSR_ERR_INVALID_ARG, "osr_pln_loss_bwd: distance_type %d", distance_type);
    OSR_REQUIRE(workspace_bytes >= osr_pln_loss_bwd_workspace_bytes(m), OSR_ERR_WORKSPACE, "osr_pln_loss_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* rows = (float*)workspace;
    int* pair_idx = (int*)((char*)workspace + 16);
    float* pair_coef = (float*)((char*)workspace + 16 + m * 8);
    float* row_inv = (float*)((char*)workspace + 16 + m * 16);
    float* pair_dist = (float*)((char*)workspace + 16 + m * 20);
    hipLaunchKernelGGL(count_rows_kernel, dim3(1), dim3(256), 0, st, (const long long*)gt_classes, (long long)m, 0, -1, rows);
    OSR_CHECK_LAUNCH("osr_pln_loss_bwd(count)");
    const float s = loss_scale * loss_weight;
    const size_t smem_rows = (size_t)kr * d * 4, smem_protos = (size_t)(kr * d + 3 * kr + d) * 4;
    if (smem_protos > 64 * 1024) {
        static osr_dev_mask attr{0};
        osr_once_per_device(attr, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pln_bwd_rows_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pln_bwd_rows_kernel<OSR_PLN_REG>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pln_bwd_protos_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        });
    }
    if (d <= 256)
        hipLaunchKernelGGL(pln_bwd_rows_kernel<4>, dim3(256), dim3(256), smem_rows, st, emb, (long long)m, d, protos_raw, num_known, reps, distance_type,
                           (const long long*)gt_classes, ious, iou_thr, alpha, beta, s, (const float*)rows, d_emb, pair_idx, pair_coef, pair_dist, row_inv);
    else
        hipLaunchKernelGGL(pln_bwd_rows_kernel<OSR_PLN_REG>, dim3(256), dim3(256), smem_rows, st, emb, (long long)m, d, protos_raw, num_known, reps, distance_type,
                           (const long long*)gt_classes, ious, iou_thr, alpha, beta, s, (const float*)rows, d_emb, pair_idx, pair_coef, pair_dist, row_inv);
    OSR_CHECK_LAUNCH("osr_pln_loss_bwd(rows)");
    hipLaunchKernelGGL(pln_bwd_protos_kernel, dim3((unsigned)kr), dim3(1024), smem_protos, st, emb, (long long)m, d, protos_raw, num_known, reps, distance_type,
                       (const int*)pair_idx, (const float*)pair_coef, (const float*)pair_dist, (const float*)row_inv, alpha, beta, s, (const float*)rows,
                       accumulate_protos, d_protos);
    OSR_CHECK_LAUNCH("osr_pln_loss_bwd(protos)");
    return OSR_OK;
}

extern "C" osr_status osr_pln_loss_bwd(const float* emb, int64_t m, int32_t d, const float* protos_raw, int32_t num_known, const int64_t* gt_classes,
                                       const float* ious, float iou_thr, float alpha, float beta, float loss_weight, float loss_scale, float* d_emb,
                                       float* d_protos, int32_t accumulate_protos, void* workspace, int64_t workspace_bytes, void* stream) {
    return osr_pln_loss_bwd_ex(emb, m, d, protos_raw, num_known, 1, OSR_DIST_COS, gt_classes, ious, iou_thr, alpha, beta, loss_weight, loss_scale, d_emb,
                               d_protos, accumulate_protos, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------
// elementwise helpers of the backward graph
// ------------------------------------------------------------------------------------------------------
template <class TG, class TA>
__global__ __launch_bounds__(256) void relu_mask_kernel(TG* __restrict__ g, const TA* __restrict__ act, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (!(osr_to_float(act[i]) > 0.f)) g[i] = osr_from_float<TG>(0.f);
}

/* g[i] = act[i] > 0 ? g[i] : 0 (in place): the gradient through a ReLU whose OUTPUT is act. */
extern "C" osr_status osr_relu_mask(void* g, int32_t g_dtype, const void* act, int32_t act_dtype, int64_t n, void* stream) {
    OSR_REQUIRE(g && act && n >= 0 && osr_dtype_ok(g_dtype) && osr_dtype_ok(act_dtype), OSR_ERR_INVALID_ARG, "osr_relu_mask: bad arguments");
    if (n == 0) return OSR_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096));
#define RM(TGt, TAt) hipLaunchKernelGGL((relu_mask_kernel<TGt, TAt>), grid, dim3(256), 0, st, (TGt*)g, (const TAt*)act, (long long)n)
    if (g_dtype == OSR_F32) { if (act_dtype == OSR_F32) RM(float, float); else if (act_dtype == OSR_F16) RM(float, f16_t); else RM(float, bf16_t); }
    else if (g_dtype == OSR_F16) { if (act_dtype == OSR_F32) RM(f16_t, float); else if (act_dtype == OSR_F16) RM(f16_t, f16_t); else RM(f16_t, bf16_t); }
    else { if (act_dtype == OSR_F32) RM(bf16_t, float); else if (act_dtype == OSR_F16) RM(bf16_t, f16_t); else RM(bf16_t, bf16_t); }
#undef RM
    OSR_CHECK_LAUNCH("osr_relu_mask");
    return OSR_OK;
}

// out[i] = a[i] (fp32, nullable) + b[i] (T, nullable), stored as T
template <class T>
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = osr_from_float<T>((a ? a[i] : 0.f) + (b ? osr_to_float(b[i]) : 0.f));
}

extern "C" osr_status osr_add_cast(const float* a_f32, const void* b, void* out, int32_t dtype, int64_t n, void* stream) {
    OSR_REQUIRE(out && (a_f32 || b) && n >= 0 && osr_dtype_ok(dtype), OSR_ERR_INVALID_ARG, "osr_add_cast: bad arguments");
    if (n == 0) return OSR_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096));
    if (dtype == OSR_F32) hipLaunchKernelGGL(add_cast_kernel<float>, grid, dim3(256), 0, st, a_f32, (const float*)b, (float*)out, (long long)n);
    else if (dtype == OSR_F16) hipLaunchKernelGGL(add_cast_kernel<f16_t>, grid, dim3(256), 0, st, a_f32, (const f16_t*)b, (f16_t*)out, (long long)n);
    else hipLaunchKernelGGL(add_cast_kernel<bf16_t>, grid, dim3(256), 0, st, a_f32, (const bf16_t*)b, (bf16_t*)out, (long long)n);
    OSR_CHECK_LAUNCH("osr_add_cast");
    return OSR_OK;
}

// FPN top-down backward: the nearest-2x upsample-add  fine = lateral + up(coarse)  sends  d coarse[y][x] += sum of the (up to)
// four fine gradients it was copied to. out = base (nullable) + sumpool2x2(fine); same dtype everywhere. stride > 0 selects the
// LastLevelMaxPool variant instead: out[y*s][x*s] = base + fine[y][x] (p6 = p5[::2, ::2]), other pixels = base.
template <class T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ fine, int hf, int wf, const T* __restrict__ base, T* __restrict__ out, int n,
                                                       int hc, int wc, int c, int mode) {
    const long long total = (long long)n * hc * wc * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long long px = i / c;
        const int x = (int)(px % wc); px /= wc;
        const int y = (int)(px % hc);
        const int img = (int)(px / hc);
        float s = base ? osr_to_float(base[i]) : 0.f;
        if (mode == 0) {  // coarse <- 2x2 of fine
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int fy = 2 * y + dy, fx = 2 * x + dx;
                    if (fy < hf && fx < wf) s += osr_to_float(fine[(((long long)img * hf + fy) * wf + fx) * c + ch]);
                }
        } else {  // out is the finer map (hc x wc); "fine" holds the subsampled one (hf x wf)
            if ((y & 1) == 0 && (x & 1) == 0 && (y >> 1) < hf && (x >> 1) < wf) s += osr_to_float(fine[(((long long)img * hf + (y >> 1)) * wf + (x >> 1)) * c + ch]);
        }
        out[i] = osr_from_float<T>(s);
    }
}

// The same for 2-byte elements and c % 8 == 0: a thread owns eight channels of one output pixel (16-byte loads and stores; the scalar
// form above moved 2 bytes per lane and divided 64-bit indices per element: 0.37 ms for the p2 -> p3 sum of a 16-image step). Same order
// of additions: base, then the fine pixels row by row.
template <class T>
__global__ __launch_bounds__(256) void pool_bwd_v8_kernel(const T* __restrict__ fine, int hf, int wf, const T* __restrict__ base, T* __restrict__ out, int n,
                                                          int hc, int wc, int c8, int mode) {
    typedef T v8 __attribute__((ext_vector_type(8)));
    const long long total = (long long)n * hc * wc * c8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % c8);
        const int px = (int)(i / c8);  // (< 2^31 pixels: checked by the caller)
        const int x = px % wc, yy = px / wc, y = yy % hc, img = yy / hc;
        float s[8];
        if (base) {
            const v8 b = reinterpret_cast<const v8*>(base)[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = (float)b[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = 0.f;
        }
        if (mode == 0) {  // coarse <- 2x2 of fine: the (up to) four loads side by side
            const int fy = 2 * y, fx = 2 * x;
            const bool okx = fx + 1 < wf, oky = fy + 1 < hf;
            const v8* f0 = reinterpret_cast<const v8*>(fine) + (((long long)img * hf + fy) * wf + fx) * c8 + cg;
            const v8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            const v8 a00 = f0[0];
            const v8 a01 = okx ? f0[c8] : zero;
            const v8 a10 = oky ? f0[(long long)wf * c8] : zero;
            const v8 a11 = (okx && oky) ? f0[(long long)wf * c8 + c8] : zero;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[e] += (float)a00[e];
                if (okx) s[e] += (float)a01[e];
                if (oky) s[e] += (float)a10[e];
                if (okx && oky) s[e] += (float)a11[e];
            }
        } else if ((y & 1) == 0 && (x & 1) == 0 && (y >> 1) < hf && (x >> 1) < wf) {
            const v8 a = reinterpret_cast<const v8*>(fine)[(((long long)img * hf + (y >> 1)) * wf + (x >> 1)) * c8 + cg];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)a[e];
        }
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = osr_from_float<T>(s[e]);
        reinterpret_cast<v8*>(out)[i] = o;
    }
}

extern "C" osr_status osr_pool_bwd(const void* src, int32_t hs, int32_t ws, const void* base, void* out, int32_t n, int32_t ho, int32_t wo, int32_t c,
                                   int32_t mode, int32_t dtype, void* stream) {
    OSR_REQUIRE(src && out && n >= 1 && hs >= 1 && ws >= 1 && ho >= 1 && wo >= 1 && c >= 1 && (mode == 0 || mode == 1) && osr_dtype_ok(dtype), OSR_ERR_INVALID_ARG,
                "osr_pool_bwd: bad arguments");
    if (mode == 0) OSR_REQUIRE(ho == (hs + 1) / 2 && wo == (ws + 1) / 2, OSR_ERR_INVALID_ARG, "osr_pool_bwd: mode 0 needs out = ceil(src / 2)");
    else OSR_REQUIRE(hs == (ho - 1) / 2 + 1 && ws == (wo - 1) / 2 + 1, OSR_ERR_INVALID_ARG, "osr_pool_bwd: mode 1 needs src = floor((out - 1) / 2) + 1");
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n * ho * wo * c;
    if (dtype != OSR_F32 && c % 8 == 0 && (long long)n * ho * wo < (1ll << 31) &&
        ((((uintptr_t)src) | ((uintptr_t)base) | ((uintptr_t)out)) & 15) == 0) {
        const long long t8 = total / 8;
        const dim3 g8((unsigned)((t8 + 255) / 256 < 16384 ? (t8 + 255) / 256 : 16384));
        if (dtype == OSR_F16) hipLaunchKernelGGL(pool_bwd_v8_kernel<f16_t>, g8, dim3(256), 0, st, (const f16_t*)src, hs, ws, (const f16_t*)base, (f16_t*)out, n, ho, wo, c / 8, mode);
        else hipLaunchKernelGGL(pool_bwd_v8_kernel<bf16_t>, g8, dim3(256), 0, st, (const bf16_t*)src, hs, ws, (const bf16_t*)base, (bf16_t*)out, n, ho, wo, c / 8, mode);
        OSR_CHECK_LAUNCH("osr_pool_bwd");
        return OSR_OK;
    }
    const dim3 grid((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096));
    if (dtype == OSR_F32) hipLaunchKernelGGL(pool_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)src, hs, ws, (const float*)base, (float*)out, n, ho, wo, c, mode);
    else if (dtype == OSR_F16) hipLaunchKernelGGL(pool_bwd_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)src, hs, ws, (const f16_t*)base, (f16_t*)out, n, ho, wo, c, mode);
    else hipLaunchKernelGGL(pool_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)src, hs, ws, (const bf16_t*)base, (bf16_t*)out, n, ho, wo, c, mode);
    OSR_CHECK_LAUNCH("osr_pool_bwd");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// SGD with momentum and weight decay ([d2] build_optimizer -> torch.optim.SGD): g' = g * grad_scale * row_scale + wd * p;
// v = mu * v + g';  p -= lr * v;  then the low-precision working copy lp = (T)(p * row_scale) in the same layout.
// row_scale (nullable, one value per leading-dimension row of row_elems elements) is the folded FrozenBN scale of a conv:
// the master weight is the un-folded one, the kernels read w * scale, and the chain rule multiplies the gradient by it.
// ------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, long long n, float lr, float mu,
                                                  float wd, float grad_scale, const float* __restrict__ row_scale, long long row_elems, T* __restrict__ lp,
                                                  const int* __restrict__ gate) {
    if (gate && *gate == 0) return;  // this iteration's gradients held an inf / NaN (osr_check_finite): leave parameters and momentum alone
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float rs = row_scale ? row_scale[i / row_elems] : 1.0f;
        float pi = p[i], vi = v[i];
        osr_sgd_element(pi, vi, g[i], rs, lr, mu, wd, grad_scale);
        v[i] = vi;
        p[i] = pi;
        if (lp) lp[i] = osr_from_float<T>(pi * rs);
    }
}

extern "C" osr_status osr_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                                   float grad_scale, const float* row_scale, int64_t row_elems, void* lowp_copy, int32_t lowp_dtype, const int32_t* apply_flag,
                                   void* stream) {
    OSR_REQUIRE(param && grad && momentum_buf && n >= 0, OSR_ERR_INVALID_ARG, "osr_sgd_step: null pointer / bad n");
    OSR_REQUIRE(!row_scale || row_elems >= 1, OSR_ERR_INVALID_ARG, "osr_sgd_step: row_elems must be >= 1 with a row scale");
    OSR_REQUIRE(!lowp_copy || osr_dtype_ok(lowp_dtype), OSR_ERR_INVALID_ARG, "osr_sgd_step: bad low-precision dtype");
    if (n == 0) return OSR_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192));
    const long long re = row_scale ? row_elems : 1;
    if (!lowp_copy || lowp_dtype == OSR_F32)
        hipLaunchKernelGGL(sgd_kernel<float>, grid, dim3(256), 0, st, param, grad, momentum_buf, (long long)n, lr, momentum, weight_decay, grad_scale, row_scale, re, (float*)lowp_copy, apply_flag);
    else if (lowp_dtype == OSR_F16)
        hipLaunchKernelGGL(sgd_kernel<f16_t>, grid, dim3(256), 0, st, param, grad, momentum_buf, (long long)n, lr, momentum, weight_decay, grad_scale, row_scale, re, (f16_t*)lowp_copy, apply_flag);
    else
        hipLaunchKernelGGL(sgd_kernel<bf16_t>, grid, dim3(256), 0, st, param, grad, momentum_buf, (long long)n, lr, momentum, weight_decay, grad_scale, row_scale, re, (bf16_t*)lowp_copy, apply_flag);
    OSR_CHECK_LAUNCH("osr_sgd_step");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// Overflow guard of the fp16 training step (static / dynamic loss scaling): *flag is cleared when any x[i] is inf or NaN.
// The caller presets *flag = 1 once per iteration, runs this over the (all-reduced) flat gradient buffer and hands the flag to
// every osr_sgd_step launch of the iteration, which then leave parameters and momentum untouched -- no host sync in between.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ x, long long n, int* __restrict__ flag) {
    bool bad = false;
    const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        bad |= !(osr_finite(v.x) && osr_finite(v.y) && osr_finite(v.z) && osr_finite(v.w));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bad |= !osr_finite(x[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 0;  // every racing writer stores the same value
}

extern "C" osr_status osr_check_finite(const float* x, int64_t n, int32_t* flag, void* stream) {
    OSR_REQUIRE(x && flag && n >= 0, OSR_ERR_INVALID_ARG, "osr_check_finite: null pointer / bad n");
    OSR_REQUIRE((((uintptr_t)x) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_check_finite: x must be 16-byte aligned");
    if (n == 0) return OSR_OK;
    const long long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks < 4096 ? blocks : 4096))), dim3(256), 0, (hipStream_t)stream, x, (long long)n,
                       flag);
    OSR_CHECK_LAUNCH("osr_check_finite");
    return OSR_OK;
}


// ------------------------------------------------------------------------------------------------------
// Backward-data weight of a convolution from its forward weight, once per SGD step (instead of flip + permute + copy kernels of
// the tensor library every iteration): dst[ci][kh-1-y][kw-1-x][co] = src[co][y][x][ci]. For 1x1 layers and FC matrices this is
// the plain transpose. 32 x 32 tiles through LDS, both sides coalesced; 2- or 4-byte elements.
// ------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const T* __restrict__ src, T* __restrict__ dst, int cout, int kh, int kw, int cin) {
    __shared__ T tile[32][33];
    const int tap = blockIdx.z, y = tap / kw, x = tap % kw;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const long long taps = (long long)kh * kw;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < cout && ci < cin) tile[r][tx] = src[((long long)co * taps + tap) * cin + ci];
    }
    __syncthreads();
    const int ftap = (kh - 1 - y) * kw + (kw - 1 - x);
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < cin && co < cout) dst[((long long)ci * taps + ftap) * cout + co] = tile[tx][r];
    }
}

extern "C" osr_status osr_pack_dgrad_weight(const void* weight, void* out, int32_t cout, int32_t kh, int32_t kw, int32_t cin, int32_t dtype, void* stream) {
    OSR_REQUIRE(weight && out && weight != out, OSR_ERR_INVALID_ARG, "osr_pack_dgrad_weight: null or aliased pointers");
    OSR_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1 && kh * kw <= 65535 && osr_dtype_ok(dtype), OSR_ERR_INVALID_ARG, "osr_pack_dgrad_weight: bad sizes / dtype");
    const dim3 grid((cin + 31) / 32, (cout + 31) / 32, kh * kw);
    OSR_REQUIRE(grid.y <= 65535, OSR_ERR_UNSUPPORTED, "osr_pack_dgrad_weight: cout too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OSR_F32) hipLaunchKernelGGL(pack_dgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)weight, (float*)out, cout, kh, kw, cin);
    else hipLaunchKernelGGL(pack_dgrad_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)weight, (unsigned short*)out, cout, kh, kw, cin);
    OSR_CHECK_LAUNCH("osr_pack_dgrad_weight");
    return OSR_OK;
}
