// Training-side targets and losses (forward) for gfx950 -- include/osr.h "train step, forward half".
//
// Reference call sites (/root/reference/openset_rcnn/modeling/):
//   proposal_generator/classification_free_rpn.py:320-411 label_and_sample_anchors, :414-491 losses
//   box_regression_w_iou.py:49-61 ("iou" box loss)            roi_heads/osrcnn_roi_heads.py:137-230
//   roi_heads/osrcnn_fast_rcnn.py:266-370                      roi_heads/prototype_learning_network.py:117-187
//   roi_heads/softmax_classifier.py:266-285
// and the [d2] primitives used there: pairwise_iou, Matcher, subsample_labels, Box2BoxTransform[Linear].get_deltas,
// add_ground_truth_to_proposals. Random sampling takes caller-supplied uniform keys (k smallest keys per class,
// ties to the lower index) instead of torch.randperm: same distribution, reproducible.
// Every reduction is two-stage (per-workgroup partials in fixed order, then one workgroup), so losses are bitwise
// reproducible run to run. Compile with -ffp-contract=off (labels / sampled index lists must be bit-exact).
#include "osr_common.h"
#include "osr_box_loss.h"
#include "osr_pln_dist.h"
static_assert(OSR_LOSS_IOU == OSR_BOX_LOSS_IOU && OSR_LOSS_SMOOTH_L1 == OSR_BOX_LOSS_SMOOTH_L1 && OSR_LOSS_GIOU == OSR_BOX_LOSS_GIOU &&
                  OSR_LOSS_DIOU == OSR_BOX_LOSS_DIOU && OSR_LOSS_CIOU == OSR_BOX_LOSS_CIOU, "osr_box_loss.h and include/osr.h number the losses alike");

struct TrLevels {
    int num_levels, num_anchors;
    int h[OSR_MAX_LEVELS], w[OSR_MAX_LEVELS], stride[OSR_MAX_LEVELS];
    long long pred_off[OSR_MAX_LEVELS];  // element offset of level l in the level-major prediction buffers
    int aoff[OSR_MAX_LEVELS + 1];        // prefix of h*w*a: index base inside an image's concatenated anchor list
    int R;
};

static bool tr_fill(const osr_rpn_levels* in, TrLevels* o) {
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

__device__ __forceinline__ float4 tr_anchor(const TrLevels& lv, const float* __restrict__ cell, int r, int* level, int* cell_idx) {
    int l = 0;
    while (l + 1 < lv.num_levels && r >= lv.aoff[l + 1]) ++l;
    const int idx = r - lv.aoff[l], A = lv.num_anchors, a = idx % A, c = idx / A;
    const float sx = (float)(c % lv.w[l]) * (float)lv.stride[l], sy = (float)(c / lv.w[l]) * (float)lv.stride[l];
    const float* ca = cell + ((long long)l * A + a) * 4;
    *level = l; *cell_idx = idx;
    return make_float4(sx + ca[0], sy + ca[1], sx + ca[2], sy + ca[3]);
}

// [d2] pairwise_iou element: inter / (a1 + a2 - inter) when inter > 0, else 0 (a1 = GT area, a2 = box area)
__device__ __forceinline__ float tr_iou(const float4 g, const float4 b) {
    const float a1 = (g.z - g.x) * (g.w - g.y), a2 = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(g.z, b.z) - fmaxf(g.x, b.x), 0.f), h = fmaxf(fminf(g.w, b.w) - fmaxf(g.y, b.y), 0.f);
    const float inter = w * h;
    return inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
}

// ------------------------------------------------------------------------------------------------------
// anchor <-> GT matching (both Matchers share the IoU matrix and its argmax)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rpn_match_pass1(TrLevels lv, const float* __restrict__ cell, const float* __restrict__ gt,
                                                       const int* __restrict__ gt_count, int gmax, int* __restrict__ matched_idx,
                                                       float* __restrict__ matched_iou, unsigned int* __restrict__ gtmax) {
    const int img = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < lv.R;  // (no early return: the whole wave takes part in the per-GT maximum below)
    int l, ci;
    const float4 b = tr_anchor(lv, cell, valid ? r : 0, &l, &ci);
    const int G = min(gt_count[img], gmax);
    float best = 0.f;
    int bi = 0;
    for (int g = 0; g < G; ++g) {
        const float4 gb = *reinterpret_cast<const float4*>(gt + ((long long)img * gmax + g) * 4);
        const float v = valid ? tr_iou(gb, b) : 0.f;
        if (g == 0 || v > best) { best = v; bi = g; }  // first maximum wins
        // best IoU of the GT over all anchors: maximum over the wave first, one atomic per wave (every lane doing its own
        // atomicMax on the handful of per-GT words serialised 11 M atomics on 128 addresses: 0.8 ms)
        unsigned int u = __float_as_uint(v);  // IoU >= 0: uint order == float order
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) u = max(u, (unsigned int)__shfl_xor((int)u, d, 64));
        if ((threadIdx.x & 63) == 0 && u != 0u) atomicMax(gtmax + (long long)img * gmax + g, u);
    }
    if (valid) {
        matched_idx[(long long)img * lv.R + r] = bi;
        matched_iou[(long long)img * lv.R + r] = best;
    }
}

__device__ __forceinline__ signed char tr_label(float v, float lo, float hi) { return v < lo ? 0 : (v < hi ? -1 : 1); }

__global__ __launch_bounds__(256) void rpn_match_pass2(TrLevels lv, const float* __restrict__ cell, const float* __restrict__ gt,
                                                       const int* __restrict__ gt_count, int gmax, const float* __restrict__ matched_iou,
                                                       const unsigned int* __restrict__ gtmax, float reg_lo, float reg_hi, float obj_lo,
                                                       float obj_hi, int low_quality, signed char* __restrict__ labels_reg,
                                                       signed char* __restrict__ labels_obj) {
    const int img = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lv.R) return;
    const int G = min(gt_count[img], gmax);
    const long long o = (long long)img * lv.R + r;
    if (G == 0) { labels_reg[o] = 0; labels_obj[o] = 0; return; }  // [d2] Matcher on an empty matrix: everything background
    const float v = matched_iou[o];
    signed char lr = tr_label(v, reg_lo, reg_hi), lo_ = tr_label(v, obj_lo, obj_hi);
    if (low_quality) {
        int l, ci;
        const float4 b = tr_anchor(lv, cell, r, &l, &ci);
        bool lq = false;
        for (int g = 0; g < G; ++g) {
            const float4 gb = *reinterpret_cast<const float4*>(gt + ((long long)img * gmax + g) * 4);
            lq |= __float_as_uint(tr_iou(gb, b)) == gtmax[(long long)img * gmax + g];  // quality == best_of_gt (incl. the 0 == 0 quirk)
        }
        if (lq) { lr = 1; lo_ = 1; }
    }
    labels_reg[o] = lr;
    labels_obj[o] = lo_;
}

extern "C" osr_status osr_rpn_match_anchors(const osr_rpn_levels* lvl, const float* cell_anchors, int32_t n, const float* gt_boxes,
                                            const int32_t* gt_count, int32_t gmax, float reg_lo, float reg_hi, float obj_lo, float obj_hi,
                                            int32_t* matched_idx, float* matched_iou, int8_t* labels_reg, int8_t* labels_obj,
                                            void* workspace, int64_t workspace_bytes, void* stream) {
    TrLevels lv;
    OSR_REQUIRE(tr_fill(lvl, &lv), OSR_ERR_INVALID_ARG, "osr_rpn_match_anchors: bad level table");
    OSR_REQUIRE(cell_anchors && gt_boxes && gt_count && matched_idx && matched_iou && labels_reg && labels_obj && workspace, OSR_ERR_INVALID_ARG,
                "osr_rpn_match_anchors: null pointer");
    OSR_REQUIRE(n >= 1 && n <= 65535 && gmax >= 1, OSR_ERR_INVALID_ARG, "osr_rpn_match_anchors: bad n / gmax");
    OSR_REQUIRE(workspace_bytes >= (int64_t)n * gmax * 4, OSR_ERR_WORKSPACE, "osr_rpn_match_anchors: workspace needs n*gmax*4 bytes");
    OSR_REQUIRE(((uintptr_t)gt_boxes & 15) == 0, OSR_ERR_INVALID_ARG, "osr_rpn_match_anchors: gt_boxes must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)n * gmax * 4, st) != hipSuccess) { osr_set_error("osr_rpn_match_anchors: memset failed"); return OSR_ERR_LAUNCH; }
    dim3 grid((lv.R + 255) / 256, n);
    hipLaunchKernelGGL(rpn_match_pass1, grid, dim3(256), 0, st, lv, cell_anchors, gt_boxes, gt_count, gmax, matched_idx, matched_iou, (unsigned int*)workspace);
    OSR_CHECK_LAUNCH("osr_rpn_match_anchors(pass1)");
    hipLaunchKernelGGL(rpn_match_pass2, grid, dim3(256), 0, st, lv, cell_anchors, gt_boxes, gt_count, gmax, matched_iou, (const unsigned int*)workspace,
                       reg_lo, reg_hi, obj_lo, obj_hi, 1, (signed char*)labels_reg, (signed char*)labels_obj);
    OSR_CHECK_LAUNCH("osr_rpn_match_anchors(pass2)");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// block-wide "k smallest keys among members" (ties: lower index), result sorted ascending by (key, index)
// ------------------------------------------------------------------------------------------------------
#define TR_THREADS 1024
#define TR_MAXK 512

__device__ __forceinline__ void tr_bitonic_desc(unsigned long long* buf, int n) {
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = buf[i], b = buf[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { buf[i] = b; buf[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// One histogram increment per lane with `pred`, wave-aggregated: keys of a narrow range share their leading byte (uniform keys in [0.5, 1):
// one exponent), so a plain LDS atomic per lane serialises 64 ways on one address. Up to four rounds elect a leader, count the lanes
// that share its bin with one ballot and add the count once; whatever is left (the well-spread later bytes) goes out as plain atomics.
__device__ __forceinline__ void tr_hist_add(int* s_hist, int bin, bool pred) {
    const int lane = threadIdx.x & 63;
    unsigned long long act = __ballot(pred);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (!act) break;  // (wave-uniform)
        const int leader = __ffsll((long long)act) - 1;
        const int b = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(pred && bin == b) & act;
        if (lane == leader) atomicAdd(&s_hist[b], __popcll(same));
        act &= ~same;
    }
    if ((act >> lane) & 1ull) atomicAdd(&s_hist[bin], 1);
}

// member(i) -> bool, keyf(i) -> float (both must be readable for every i < cnt). Selects min(k, #members) entries; s_sel[j] low 32 bits =
// 0xffffffff - index, in ascending (key, index) order. Returns the number selected (uniform). Must be called by the whole block.
// Round 5: every pass over the list keeps TR_UNROLL independent (member, key) loads per thread in flight -- the passes used to issue one
// dependent 1-byte + 4-byte load pair per trip, ~15 passes x 88 trips of memory latency per image -- and the compaction ranks its candidates
// with ONE block scan over per-thread counts of a blocked layout (thread t owns a contiguous run: thread order = index order, which is what
// the tie rule needs) instead of one block scan per 1024 elements. Same selection bit for bit (tests/test_train_fwd.py, golden fixtures).
template <int TR_UNROLL = 1, class MemberF, class KeyF>  // TR_UNROLL: (member, key) pairs a thread keeps in flight (8 for the 89 523-anchor lists; 1 where the accessors are heavy)
__device__ int tr_select_smallest(MemberF member, KeyF keyf, int cnt, int k, unsigned long long* s_sel,
                                  int* s_hist, int* s_scan, unsigned int* s_bc) {
    const int tid = threadIdx.x, nt = blockDim.x;
    // batched walk, interleaved layout (coalesced): BODY(i, mem, v) for every i < cnt with mem = member(i), v = inverted monotone key
#define TR_WALK(BODY)                                                                                          \
    for (int i0_ = tid; i0_ < cnt; i0_ += nt * TR_UNROLL) {                                                     \
        bool mem_[TR_UNROLL];                                                                                   \
        unsigned int v_[TR_UNROLL];                                                                             \
        _Pragma("unroll") for (int u_ = 0; u_ < TR_UNROLL; ++u_) {                                              \
            const int i_ = i0_ + u_ * nt;                                                                       \
            const int ic_ = i_ < cnt ? i_ : 0;                                                                  \
            mem_[u_] = member(ic_) && i_ < cnt;                                                                 \
            v_[u_] = ~osr_float_key(keyf(ic_));                                                                 \
        }                                                                                                       \
        _Pragma("unroll") for (int u_ = 0; u_ < TR_UNROLL; ++u_) { const int i = i0_ + u_ * nt; const bool mem = mem_[u_]; const unsigned int v = v_[u_]; (void)i; (void)v; BODY }  \
    }
    // members
    int c = 0;
    TR_WALK({ c += mem ? 1 : 0; })
    int m;
    osr_block_excl_scan(c, s_scan, &m);
    if (k > m) k = m;
    if (k > TR_MAXK) k = TR_MAXK;
    if (k <= 0) return 0;
    // inverted monotone key: the k smallest floats are the k largest v
    unsigned int prefix = 0, mask = 0;
    int remaining = k;
    const bool all = m <= k;
    if (!all) {
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += nt) s_hist[i] = 0;
            __syncthreads();
            TR_WALK({ tr_hist_add(s_hist, (int)((v >> shift) & 255), mem && (v & mask) == prefix); })
            __syncthreads();
            if (tid < 64) {
                // the digit d (from 255 down) at which the count of larger digits first reaches `remaining`: lane l owns digits 255 - 4 l .. 252 - 4 l
                // (a serial walk by one thread was 256 dependent LDS reads per pass: ~12 us, eight times per call)
                const int h0 = s_hist[255 - 4 * tid], h1 = s_hist[254 - 4 * tid], h2 = s_hist[253 - 4 * tid], h3 = s_hist[252 - 4 * tid];
                const int mine = h0 + h1 + h2 + h3;
                const int before = osr_wave_incl_scan(mine) - mine;  // digits larger than this lane's four
                // first lane (lowest index = largest digits) whose inclusive count reaches `remaining`; digit 0 takes whatever is left
                const bool hit = before + mine >= remaining;
                const unsigned long long hits = __ballot(hit);
                const int sel = hits ? __ffsll((long long)hits) - 1 : 63;
                if (tid == sel) {
                    int acc = before, d = 255 - 4 * tid;
                    const int hh[4] = {h0, h1, h2, h3};
                    int j = 0;
                    for (; j < 3; ++j) {
                        if (d - j == 0 || acc + hh[j] >= remaining) break;
                        acc += hh[j];
                    }
                    s_bc[0] = prefix | ((unsigned int)(d - j) << shift);
                    s_bc[1] = (unsigned int)(remaining - acc);
                }
            }
            __syncthreads();
            prefix = s_bc[0];
            remaining = (int)s_bc[1];
            mask |= 255u << shift;
            __syncthreads();
        }
    }
    const unsigned int T = prefix;
    // compaction, blocked layout: thread t owns [t * chunk, (t + 1) * chunk): its candidates greater than T and equal to T, counted, ranked by
    // one block scan, then written in index order (candidates equal to T: the first `remaining` of them by index)
    const int chunk = (cnt + nt - 1) / nt, lo = min(tid * chunk, cnt), hi = min(lo + chunk, cnt);
#define TR_WALK_BLOCKED(BODY)                                                                                   \
    for (int i0_ = lo; i0_ < hi; i0_ += TR_UNROLL) {                                                            \
        bool mem_[TR_UNROLL];                                                                                   \
        unsigned int v_[TR_UNROLL];                                                                             \
        _Pragma("unroll") for (int u_ = 0; u_ < TR_UNROLL; ++u_) {                                              \
            const int i_ = i0_ + u_;                                                                            \
            const int ic_ = i_ < hi ? i_ : lo;                                                                  \
            mem_[u_] = member(ic_) && i_ < hi;                                                                  \
            v_[u_] = ~osr_float_key(keyf(ic_));                                                                 \
        }                                                                                                       \
        _Pragma("unroll") for (int u_ = 0; u_ < TR_UNROLL; ++u_) { const int i = i0_ + u_; const bool mem = mem_[u_]; const unsigned int v = v_[u_]; (void)i; BODY }  \
    }
    int gt = 0, eq = 0;
    TR_WALK_BLOCKED({ if (mem) { if (all || v > T) ++gt; else if (v == T) ++eq; } })
    // One packed scan: gt in bits 0..10 (the candidates above the threshold number fewer than k <= TR_MAXK in total), eq -- SATURATED at
    // TR_MAXK per thread -- above. The count of candidates EQUAL to the threshold is unbounded (all-equal keys: the whole list), so the raw
    // count would run over the packed field; only ranks below `remaining` <= TR_MAXK are ever used: while a thread's true prefix is below
    // `remaining` no earlier thread was saturated and the saturated prefix IS the true one, and once one was, the prefix is >= TR_MAXK >=
    // remaining as the true one is. 1024 threads x 512 << 11 = 2^30: the packed sum stays a positive int.
    static_assert(TR_MAXK <= 512 && TR_THREADS <= 1024, "packed (gt, eq) scan: TR_THREADS * TR_MAXK << 11 must stay below 2^31");
    int tot;
    const int packed = osr_block_excl_scan(gt | (min(eq, TR_MAXK) << 11), s_scan, &tot);
    int pg = packed & 0x7ff, pe = packed >> 11;
    const int ngt_total = tot & 0x7ff;
    TR_WALK_BLOCKED({
        if (mem) {
            const unsigned long long comp = ((unsigned long long)v << 32) | (unsigned int)(0xffffffffu - (unsigned int)i);
            if (all || v > T) { if (pg < TR_MAXK) s_sel[pg] = comp; ++pg; }
            else if (v == T) { if (pe < remaining && ngt_total + pe < TR_MAXK) s_sel[ngt_total + pe] = comp; ++pe; }
        }
    })
#undef TR_WALK
#undef TR_WALK_BLOCKED
    int kp = 1;
    while (kp < k) kp <<= 1;
    for (int i = k + tid; i < kp; i += blockDim.x) s_sel[i] = 0ull;
    __syncthreads();
    tr_bitonic_desc(s_sel, kp);  // v descending == key ascending; index ascending inside ties
    return k;
}

// [d2] subsample_labels + ClsFreeRPN._subsample_labels (classification_free_rpn.py:299-316): grid (image, which)
__global__ __launch_bounds__(TR_THREADS) void subsample_kernel(signed char* __restrict__ labels, const float* __restrict__ keys, long long R,
                                                               int num_samples, float pos_fraction, int* __restrict__ num_pos_out,
                                                               int* __restrict__ num_neg_out) {
    __shared__ unsigned long long s_sel[TR_MAXK];
    __shared__ int s_pos[TR_MAXK], s_neg[TR_MAXK];
    __shared__ int s_hist[256], s_scan[32];
    __shared__ unsigned int s_bc[2];
    const int img = blockIdx.x, tid = threadIdx.x;
    signed char* lab = labels + (long long)img * R;
    const float* ky = keys + (long long)img * R;
    const int cnt = (int)R;
    const int want_pos = (int)((float)num_samples * pos_fraction);
    const int np = tr_select_smallest<8>([&](int i) { return lab[i] == 1; }, [&](int i) { return ky[i]; }, cnt, want_pos, s_sel, s_hist, s_scan, s_bc);
    for (int j = tid; j < np; j += blockDim.x) s_pos[j] = (int)(0xffffffffu - (unsigned int)(s_sel[j] & 0xffffffffull));
    __syncthreads();
    const int nn = tr_select_smallest<8>([&](int i) { return lab[i] == 0; }, [&](int i) { return ky[i]; }, cnt, num_samples - np, s_sel, s_hist, s_scan, s_bc);
    for (int j = tid; j < nn; j += blockDim.x) s_neg[j] = (int)(0xffffffffu - (unsigned int)(s_sel[j] & 0xffffffffull));
    __syncthreads();
    for (int i = tid; i < cnt; i += blockDim.x) lab[i] = -1;  // label.fill_(-1)
    __syncthreads();
    for (int j = tid; j < np; j += blockDim.x) lab[s_pos[j]] = 1;  // scatter_(pos_idx, 1)
    for (int j = tid; j < nn; j += blockDim.x) lab[s_neg[j]] = 0;  // scatter_(neg_idx, 0)
    if (tid == 0) { num_pos_out[img] = np; num_neg_out[img] = nn; }
}

extern "C" osr_status osr_subsample_labels(int8_t* labels, const float* keys, int32_t n, int64_t r, int32_t num_samples, float positive_fraction,
                                           int32_t* num_pos_out, int32_t* num_neg_out, void* stream) {
    OSR_REQUIRE(labels && keys && num_pos_out && num_neg_out, OSR_ERR_INVALID_ARG, "osr_subsample_labels: null pointer");
    OSR_REQUIRE(n >= 1 && r >= 1 && r < (1ll << 30), OSR_ERR_INVALID_ARG, "osr_subsample_labels: bad sizes");
    OSR_REQUIRE(num_samples >= 1 && num_samples <= TR_MAXK && positive_fraction >= 0.f && positive_fraction <= 1.f, OSR_ERR_UNSUPPORTED,
                "osr_subsample_labels: num_samples must be in 1..%d", TR_MAXK);
    hipLaunchKernelGGL(subsample_kernel, dim3(n), dim3(TR_THREADS), 0, (hipStream_t)stream, (signed char*)labels, keys, (long long)r, num_samples,
                       positive_fraction, num_pos_out, num_neg_out);
    OSR_CHECK_LAUNCH("osr_subsample_labels");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// matched GT boxes + centerness targets (classification_free_rpn.py:386-402)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rpn_targets_kernel(TrLevels lv, const float* __restrict__ cell, const float* __restrict__ gt,
                                                          const int* __restrict__ gt_count, int gmax, const int* __restrict__ matched_idx,
                                                          const signed char* __restrict__ labels_obj, float* __restrict__ matched_boxes,
                                                          float* __restrict__ ctr_target) {
    const int img = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lv.R) return;
    const long long o = (long long)img * lv.R + r;
    const int G = min(gt_count[img], gmax);
    if (G == 0) {
        *reinterpret_cast<float4*>(matched_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        ctr_target[o] = 0.f;
        return;
    }
    int l, ci;
    const float4 a = tr_anchor(lv, cell, r, &l, &ci);
    const float4 g = *reinterpret_cast<const float4*>(gt + ((long long)img * gmax + matched_idx[o]) * 4);
    *reinterpret_cast<float4*>(matched_boxes + o * 4) = g;
    // [d2] Box2BoxTransformLinear(normalize_by_size=True).get_deltas, reordered to l, r, t, b
    const float cx = 0.5f * (a.x + a.z), cy = 0.5f * (a.y + a.w), sw = a.z - a.x, sh = a.w - a.y;
    float dl = (cx - g.x) / sw, dt = (cy - g.y) / sh, dr = (g.z - cx) / sw, db = (g.w - cy) / sh;
    if (!(dl >= 0.f && dr >= 0.f && dt >= 0.f && db >= 0.f)) { dl = 0.f; dr = 0.f; dt = 0.f; db = 0.f; }
    float c = sqrtf((fminf(dl, dr) / (fmaxf(dl, dr) + 1e-12f)) * (fminf(dt, db) / (fmaxf(dt, db) + 1e-12f)));
    if (labels_obj[o] == 0) c = 0.f;
    ctr_target[o] = c;
}

extern "C" osr_status osr_rpn_anchor_targets(const osr_rpn_levels* lvl, const float* cell_anchors, int32_t n, const float* gt_boxes,
                                             const int32_t* gt_count, int32_t gmax, const int32_t* matched_idx, const int8_t* labels_obj,
                                             float* matched_boxes, float* ctr_target, void* stream) {
    TrLevels lv;
    OSR_REQUIRE(tr_fill(lvl, &lv), OSR_ERR_INVALID_ARG, "osr_rpn_anchor_targets: bad level table");
    OSR_REQUIRE(cell_anchors && gt_boxes && gt_count && matched_idx && labels_obj && matched_boxes && ctr_target, OSR_ERR_INVALID_ARG,
                "osr_rpn_anchor_targets: null pointer");
    OSR_REQUIRE(n >= 1 && n <= 65535 && gmax >= 1, OSR_ERR_INVALID_ARG, "osr_rpn_anchor_targets: bad n / gmax");
    OSR_REQUIRE((((uintptr_t)gt_boxes | (uintptr_t)matched_boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_rpn_anchor_targets: box arrays must be 16-byte aligned");
    hipLaunchKernelGGL(rpn_targets_kernel, dim3((lv.R + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, lv, cell_anchors, gt_boxes, gt_count, gmax,
                       matched_idx, (const signed char*)labels_obj, matched_boxes, ctr_target);
    OSR_CHECK_LAUNCH("osr_rpn_anchor_targets");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// deterministic two-stage reductions
// ------------------------------------------------------------------------------------------------------
#define RED_BLOCKS 256
#define RED_MAXV 8

template <int NV>
__device__ __forceinline__ void tr_block_reduce_store(float v[NV], float* __restrict__ partial /* [gridDim.x][NV] */) {
    __shared__ float s_red[RED_MAXV][16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        float x = v[q];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
        if (lane == 0) s_red[q][wid] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float x = 0.f;
        for (int w = 0; w < nw; ++w) x += s_red[threadIdx.x][w];
        partial[(long long)blockIdx.x * NV + threadIdx.x] = x;
    }
}

// out[q] = scale[q] * sum_b partial[b][q]   (fixed order)
struct TrScale { float s[RED_MAXV]; };
// out[q] = scale[q] * sum_b partial[b][q] (fixed order); with norm_col >= 0 every column q < norm_col is also divided by
// max(sum_b partial[b][norm_col], 1) -- the "rows that count" normaliser when the row list is padded.
// One wave: lane l adds the partials of workgroups l, l + 64, ... of column q, then the 64 lane sums go down a shuffle tree -- a fixed
// order, so the value is reproducible. (One thread walking all the workgroups' partials was a 16-30 us dependent chain per loss, four
// times on the training step's critical stream.)
__device__ __forceinline__ float tr_wave_column_sum(const float* __restrict__ partial, int nblocks, int nv, int q) {
    const int lane = threadIdx.x & 63;
    float x = 0.f;
#pragma unroll 4
    for (int b = lane; b < nblocks; b += 64) x += partial[(long long)b * nv + q];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    return __shfl(x, 0, 64);
}

__global__ __launch_bounds__(64) void tr_final_reduce(const float* __restrict__ partial, int nblocks, int nv, TrScale scale, int norm_col, float* __restrict__ out) {
    const float c = norm_col >= 0 ? tr_wave_column_sum(partial, nblocks, nv, norm_col) : 1.0f;
    for (int q = 0; q < nv; ++q) {
        float x = tr_wave_column_sum(partial, nblocks, nv, q);
        if (norm_col >= 0 && q < norm_col) x = x / fmaxf(c, 1.0f);
        if (threadIdx.x == 0) out[q] = x * scale.s[q];
    }
}

// ------------------------------------------------------------------------------------------------------
// RPN losses forward: "iou" localisation loss + L1 centerness loss (+ the four logged anchor counts)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rpn_losses_kernel(TrLevels lv, const float* __restrict__ cell, int n, const float* __restrict__ pred_deltas,
                                                         const float* __restrict__ pred_ctr, const signed char* __restrict__ labels_reg,
                                                         const signed char* __restrict__ labels_obj, const float* __restrict__ matched_boxes,
                                                         const float* __restrict__ ctr_target, int box_type, float box_beta, float ctr_beta,
                                                         float* __restrict__ partial) {
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // loc, ctr, num_pos, num_neg, obj_pos, obj_neg
    const long long total = (long long)n * lv.R;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(i / lv.R), r = (int)(i - (long long)img * lv.R);
        const signed char lr = labels_reg[i], lo = labels_obj[i];
        v[2] += lr == 1; v[3] += lr == 0; v[4] += lo == 1; v[5] += lo == 0;
        if (lr != 1 && lo == -1) continue;
        int l, ci;
        const float4 a = tr_anchor(lv, cell, r, &l, &ci);
        const long long pi = lv.pred_off[l] + (long long)img * (lv.aoff[l + 1] - lv.aoff[l]) + ci;
        if (lr == 1) {
            const float4 d = *reinterpret_cast<const float4*>(pred_deltas + pi * 4);
            const float cx = 0.5f * (a.x + a.z), cy = 0.5f * (a.y + a.w), aw = a.z - a.x, ah = a.w - a.y;
            const float4 g = *reinterpret_cast<const float4*>(matched_boxes + i * 4);
            if (box_type == OSR_LOSS_SMOOTH_L1) {
                // the raw deltas against Box2BoxTransformLinear.get_deltas(anchor, gt) (box_regression_w_iou.py:40-48)
                v[0] += osr_smooth_l1(d.x - (cx - g.x) / aw, box_beta) + osr_smooth_l1(d.y - (cy - g.y) / ah, box_beta) +
                        osr_smooth_l1(d.z - (g.z - cx) / aw, box_beta) + osr_smooth_l1(d.w - (g.w - cy) / ah, box_beta);
            } else {
                // the decoded box (apply_deltas: ReLU of the deltas) against the matched box; "iou": diag(pairwise_iou(pred, gt)).clamp(min=1e-6),
                // pred in the "b1" role (box_regression_w_iou.py:49-61); giou / diou / ciou: :62-82
                const float4 pb = make_float4(cx - fmaxf(d.x, 0.f) * aw, cy - fmaxf(d.y, 0.f) * ah, cx + fmaxf(d.z, 0.f) * aw, cy + fmaxf(d.w, 0.f) * ah);
                float unused[4];
                v[0] += osr_box_loss<false>(box_type, pb, g, unused);
            }
        }
        if (lo != -1) v[1] += osr_smooth_l1(pred_ctr[pi] - ctr_target[i], ctr_beta);
    }
    tr_block_reduce_store<6>(v, partial);
}

extern "C" osr_status osr_rpn_losses_fwd_ex(const osr_rpn_levels* lvl, const float* cell_anchors, int32_t n, const float* pred_deltas,
                                         const float* pred_ctr, const int8_t* labels_reg, const int8_t* labels_obj, const float* matched_boxes,
                                         const float* ctr_target, float loc_weight, float ctr_weight, int32_t batch_size_per_image,
                                            const osr_loss_options* opt, float* out6, void* workspace, int64_t workspace_bytes, void* stream) {
    const int box_type = opt ? opt->box_loss_type : OSR_LOSS_IOU;
    const float box_beta = opt ? opt->box_smooth_l1_beta : 0.f, ctr_beta = opt ? opt->aux_smooth_l1_beta : 0.f;
    OSR_REQUIRE(box_type >= OSR_LOSS_IOU && box_type <= OSR_LOSS_CIOU && box_beta >= 0.f && ctr_beta >= 0.f, OSR_ERR_INVALID_ARG,
                "osr_rpn_losses_fwd: bad loss options (type %d)", box_type);
    TrLevels lv;
    OSR_REQUIRE(tr_fill(lvl, &lv), OSR_ERR_INVALID_ARG, "osr_rpn_losses_fwd: bad level table");
    OSR_REQUIRE(cell_anchors && pred_deltas && pred_ctr && labels_reg && labels_obj && matched_boxes && ctr_target && out6 && workspace,
                OSR_ERR_INVALID_ARG, "osr_rpn_losses_fwd: null pointer");
    OSR_REQUIRE(n >= 1 && batch_size_per_image >= 1, OSR_ERR_INVALID_ARG, "osr_rpn_losses_fwd: bad n / batch size");
    OSR_REQUIRE(workspace_bytes >= (int64_t)RED_BLOCKS * 6 * 4, OSR_ERR_WORKSPACE, "osr_rpn_losses_fwd: workspace needs %d bytes", RED_BLOCKS * 6 * 4);
    OSR_REQUIRE((((uintptr_t)pred_deltas | (uintptr_t)matched_boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_rpn_losses_fwd: box arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const float norm = (float)batch_size_per_image * (float)n;
    const TrScale scale = {{1.0f / norm * loc_weight, 1.0f / norm * ctr_weight, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f}};
    hipLaunchKernelGGL(rpn_losses_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, lv, cell_anchors, n, pred_deltas, pred_ctr, (const signed char*)labels_reg,
                       (const signed char*)labels_obj, matched_boxes, ctr_target, box_type, box_beta, ctr_beta, partial);
    OSR_CHECK_LAUNCH("osr_rpn_losses_fwd");
    hipLaunchKernelGGL(tr_final_reduce, dim3(1), dim3(64), 0, st, partial, RED_BLOCKS, 6, scale, -1, out6);
    OSR_CHECK_LAUNCH("osr_rpn_losses_fwd(final)");
    return OSR_OK;
}

extern "C" osr_status osr_rpn_losses_fwd(const osr_rpn_levels* lvl, const float* cell_anchors, int32_t n, const float* pred_deltas,
                                         const float* pred_ctr, const int8_t* labels_reg, const int8_t* labels_obj, const float* matched_boxes,
                                         const float* ctr_target, float loc_weight, float ctr_weight, int32_t batch_size_per_image, float* out6,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
    return osr_rpn_losses_fwd_ex(lvl, cell_anchors, n, pred_deltas, pred_ctr, labels_reg, labels_obj, matched_boxes, ctr_target, loc_weight, ctr_weight,
                                 batch_size_per_image, nullptr, out6, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------
// RoI heads: append GT, match, sample (osrcnn_roi_heads.py:177-216): one workgroup per image
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void roi_match_sample_kernel(const float* __restrict__ prop_boxes, const float* __restrict__ prop_logits,
                                                                      const int* __restrict__ prop_count, long long pcap, const float* __restrict__ gt,
                                                                      const long long* __restrict__ gt_classes, const int* __restrict__ gt_count,
                                                                      int gmax, const float* __restrict__ keys, int num_classes, int batch_size,
                                                                      float pos_fraction, float iou_thr, float gt_logit, int* __restrict__ ws_cls,
                                                                      float* __restrict__ ws_iou, int* __restrict__ ws_midx, float* __restrict__ out_boxes,
                                                                      float* __restrict__ out_logits, long long* __restrict__ out_cls,
                                                                      float* __restrict__ out_iou, float* __restrict__ out_gt, int* __restrict__ out_src,
                                                                      int* __restrict__ out_bidx, int* __restrict__ out_counts) {
    __shared__ unsigned long long s_sel[TR_MAXK];
    __shared__ int s_idx[TR_MAXK];
    __shared__ int s_hist[256], s_scan[32];
    __shared__ unsigned int s_bc[2];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int P = min((long long)prop_count[img], pcap), G = min(gt_count[img], gmax);
    const int C = P + G;
    const long long cstride = pcap + gmax;
    int* cls = ws_cls + img * cstride;
    float* miou = ws_iou + img * cstride;
    int* midx = ws_midx + img * cstride;
    const float* ky = keys + img * cstride;
    auto keyf = [&](int j) { return ky[j < P ? j : (int)pcap + (j - P)]; };  // keys are laid out (pcap proposals, gmax GT)
    auto cand_box = [&](int j) {
        return j < P ? *reinterpret_cast<const float4*>(prop_boxes + ((long long)img * pcap + j) * 4)
                     : *reinterpret_cast<const float4*>(gt + ((long long)img * gmax + (j - P)) * 4);
    };
    for (int j = tid; j < C; j += blockDim.x) {
        const float4 b = cand_box(j);
        float best = 0.f;
        int bi = 0;
        for (int g = 0; g < G; ++g) {
            const float v = tr_iou(*reinterpret_cast<const float4*>(gt + ((long long)img * gmax + g) * 4), b);
            if (g == 0 || v > best) { best = v; bi = g; }
        }
        const bool fg = G > 0 && best >= iou_thr;  // [d2] Matcher([0.5], [0, 1])
        cls[j] = fg ? (int)gt_classes[(long long)img * gmax + bi] : num_classes;
        miou[j] = G > 0 ? best : 0.f;
        midx[j] = bi;
    }
    __syncthreads();
    // [d2] subsample_labels(gt_classes, batch_size, positive_fraction, bg = num_classes): fg first, then bg
    const int want_fg = (int)((float)batch_size * pos_fraction);
    int nsel = 0;
    const int nfg = tr_select_smallest([&](int i) { return cls[i] != num_classes && cls[i] != -1; }, keyf, C, want_fg, s_sel, s_hist, s_scan, s_bc);
    for (int j = tid; j < nfg; j += blockDim.x) s_idx[j] = (int)(0xffffffffu - (unsigned int)(s_sel[j] & 0xffffffffull));
    __syncthreads();
    nsel = nfg;
    const int nbg = tr_select_smallest([&](int i) { return cls[i] == num_classes; }, keyf, C, batch_size - nfg, s_sel, s_hist, s_scan, s_bc);
    for (int j = tid; j < nbg; j += blockDim.x) s_idx[nsel + j] = (int)(0xffffffffu - (unsigned int)(s_sel[j] & 0xffffffffull));
    __syncthreads();
    nsel += nbg;
    for (int j = tid; j < batch_size; j += blockDim.x) {
        const long long o = (long long)img * batch_size + j;
        if (j < nsel) {
            const int s = s_idx[j];
            *reinterpret_cast<float4*>(out_boxes + o * 4) = cand_box(s);
            out_logits[o] = s < P ? prop_logits[(long long)img * pcap + s] : gt_logit;
            out_cls[o] = cls[s];
            out_iou[o] = miou[s];
            *reinterpret_cast<float4*>(out_gt + o * 4) = G > 0 ? *reinterpret_cast<const float4*>(gt + ((long long)img * gmax + midx[s]) * 4)
                                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            out_src[o] = s;
            out_bidx[o] = img;
        } else {
            *reinterpret_cast<float4*>(out_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(out_gt + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            out_logits[o] = 0.f; out_cls[o] = -1; out_iou[o] = 0.f; out_src[o] = -1; out_bidx[o] = -1;
        }
    }
    if (tid == 0) { out_counts[img * 3 + 0] = nsel; out_counts[img * 3 + 1] = nfg; out_counts[img * 3 + 2] = nbg; }
}

extern "C" int64_t osr_roi_match_sample_workspace_bytes(int32_t n, int64_t pcap, int32_t gmax) {
    if (n < 1 || pcap < 1 || gmax < 1) { osr_set_error("osr_roi_match_sample_workspace_bytes: bad arguments"); return OSR_ERR_INVALID_ARG; }
    return (int64_t)n * (pcap + gmax) * 12;
}

extern "C" osr_status osr_roi_match_and_sample(const float* prop_boxes, const float* prop_logits, const int32_t* prop_count, int64_t pcap,
                                               const float* gt_boxes, const int64_t* gt_classes, const int32_t* gt_count, int32_t gmax, int32_t n,
                                               const float* keys, int32_t num_classes, int32_t batch_size, float positive_fraction, float iou_thr,
                                               float* out_boxes, float* out_logits, int64_t* out_classes, float* out_ious, float* out_gt_boxes,
                                               int32_t* out_src, int32_t* out_batch_idx, int32_t* out_counts, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(prop_boxes && prop_logits && prop_count && gt_boxes && gt_classes && gt_count && keys && out_boxes && out_logits && out_classes &&
                    out_ious && out_gt_boxes && out_src && out_batch_idx && out_counts && workspace, OSR_ERR_INVALID_ARG, "osr_roi_match_and_sample: null pointer");
    OSR_REQUIRE(n >= 1 && pcap >= 1 && gmax >= 1 && pcap + gmax < (1ll << 30), OSR_ERR_INVALID_ARG, "osr_roi_match_and_sample: bad sizes");
    OSR_REQUIRE(batch_size >= 1 && batch_size <= TR_MAXK, OSR_ERR_UNSUPPORTED, "osr_roi_match_and_sample: batch_size must be in 1..%d", TR_MAXK);
    const int64_t need = (int64_t)n * (pcap + gmax) * 12;
    OSR_REQUIRE(workspace_bytes >= need, OSR_ERR_WORKSPACE, "osr_roi_match_and_sample: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    OSR_REQUIRE((((uintptr_t)prop_boxes | (uintptr_t)gt_boxes | (uintptr_t)out_boxes | (uintptr_t)out_gt_boxes) & 15) == 0, OSR_ERR_INVALID_ARG,
                "osr_roi_match_and_sample: box arrays must be 16-byte aligned");
    char* ws = (char*)workspace;
    const int64_t c = (int64_t)n * (pcap + gmax);
    // add_ground_truth_to_proposals: logit of a GT box = log((1 - 1e-10) / (1 - (1 - 1e-10))) evaluated in double, then fp32
    const double gt_prob = 1.0 - 1e-10;
    const float gt_logit = (float)log(gt_prob / (1.0 - gt_prob));
    hipLaunchKernelGGL(roi_match_sample_kernel, dim3(n), dim3(TR_THREADS), 0, (hipStream_t)stream, prop_boxes, prop_logits, prop_count, (long long)pcap,
                       gt_boxes, (const long long*)gt_classes, gt_count, gmax, keys, num_classes, batch_size, positive_fraction, iou_thr, gt_logit,
                       (int*)ws, (float*)(ws + c * 4), (int*)(ws + c * 8), out_boxes, out_logits, (long long*)out_classes, out_ious, out_gt_boxes,
                       out_src, out_batch_idx, out_counts);
    OSR_CHECK_LAUNCH("osr_roi_match_and_sample");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// box / IoU regression losses forward (osrcnn_fast_rcnn.py:312-370)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_box_losses_kernel(const float* __restrict__ pred_deltas, int delta_stride, const float* __restrict__ pred_iou,
                                                             int iou_stride, int iou_is_logit, const float* __restrict__ prop,
                                                             const float* __restrict__ gtb, const long long* __restrict__ cls,
                                                             const float* __restrict__ gt_iou, long long m, int num_classes, float wx, float wy, float ww,
                                                             float wh, int box_type, float box_beta, float iou_beta, float* __restrict__ partial) {
    float v[3] = {0.f, 0.f, 0.f};  // box loss, IoU loss, rows that count (class >= 0; padding rows carry -1)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const long long c = cls[i];
        if (c < 0) continue;
        v[2] += 1.f;
        if (c >= num_classes) continue;
        const float4 s = *reinterpret_cast<const float4*>(prop + i * 4), t = *reinterpret_cast<const float4*>(gtb + i * 4);
        const float* d = pred_deltas + i * delta_stride;
        // [d2] Box2BoxTransform.get_deltas
        const float sw = s.z - s.x, sh = s.w - s.y, scx = s.x + 0.5f * sw, scy = s.y + 0.5f * sh;
        const float tw = t.z - t.x, th = t.w - t.y, tcx = t.x + 0.5f * tw, tcy = t.y + 0.5f * th;
        if (box_type == OSR_LOSS_SMOOTH_L1) {
            const float dx = wx * (tcx - scx) / sw, dy = wy * (tcy - scy) / sh, dw = ww * logf(tw / sw), dh = wh * logf(th / sh);
            v[0] += osr_smooth_l1(d[0] - dx, box_beta) + osr_smooth_l1(d[1] - dy, box_beta) + osr_smooth_l1(d[2] - dw, box_beta) +
                    osr_smooth_l1(d[3] - dh, box_beta);
        } else {  // [d2] Box2BoxTransform.apply_deltas, then the box loss (box_regression_w_iou.py:49-82)
            const float kClamp = 4.135166556742356f;  // log(1000 / 16)
            const float pw = expf(fminf(d[2] / ww, kClamp)) * sw, ph = expf(fminf(d[3] / wh, kClamp)) * sh;
            const float pcx = d[0] / wx * sw + scx, pcy = d[1] / wy * sh + scy;
            float unused[4];
            v[0] += osr_box_loss<false>(box_type, make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph), t, unused);
        }
        float pi = pred_iou[i * iou_stride];
        if (iou_is_logit) pi = 1.0f / (1.0f + expf(-pi));  // OpensetFastRCNNOutputLayers.forward: iou_pred(x).sigmoid()
        v[1] += osr_smooth_l1(pi - gt_iou[i], iou_beta);
    }
    tr_block_reduce_store<3>(v, partial);
}

extern "C" osr_status osr_roi_box_losses_fwd_ex(const float* pred_deltas, int32_t delta_stride, const float* pred_iou, int32_t iou_stride,
                                             int32_t iou_is_logit, const float* proposal_boxes, const float* gt_boxes, const int64_t* gt_classes,
                                             const float* gt_iou, int64_t m, int32_t num_classes, const float reg_weights[4], float box_weight,
                                                float iou_weight, const osr_loss_options* opt, float* out3, void* workspace, int64_t workspace_bytes,
                                                void* stream) {
    const int box_type = opt ? opt->box_loss_type : OSR_LOSS_SMOOTH_L1;
    const float box_beta = opt ? opt->box_smooth_l1_beta : 0.f, iou_beta = opt ? opt->aux_smooth_l1_beta : 0.f;
    OSR_REQUIRE(box_type >= OSR_LOSS_IOU && box_type <= OSR_LOSS_CIOU && box_beta >= 0.f && iou_beta >= 0.f, OSR_ERR_INVALID_ARG,
                "osr_roi_box_losses_fwd: bad loss options (type %d)", box_type);
    OSR_REQUIRE(pred_deltas && pred_iou && proposal_boxes && gt_boxes && gt_classes && gt_iou && reg_weights && out3 && workspace, OSR_ERR_INVALID_ARG,
                "osr_roi_box_losses_fwd: null pointer");
    OSR_REQUIRE(m >= 0 && delta_stride >= 4 && iou_stride >= 1, OSR_ERR_INVALID_ARG, "osr_roi_box_losses_fwd: bad m / strides");
    OSR_REQUIRE(workspace_bytes >= (int64_t)RED_BLOCKS * 3 * 4, OSR_ERR_WORKSPACE, "osr_roi_box_losses_fwd: workspace needs %d bytes", RED_BLOCKS * 3 * 4);
    OSR_REQUIRE((((uintptr_t)proposal_boxes | (uintptr_t)gt_boxes) & 15) == 0, OSR_ERR_INVALID_ARG, "osr_roi_box_losses_fwd: box arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const TrScale scale = {{box_weight, iou_weight, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    hipLaunchKernelGGL(roi_box_losses_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, pred_deltas, delta_stride, pred_iou, iou_stride, iou_is_logit, proposal_boxes,
                       gt_boxes, (const long long*)gt_classes, gt_iou, (long long)m, num_classes, reg_weights[0], reg_weights[1], reg_weights[2],
                       reg_weights[3], box_type, box_beta, iou_beta, partial);
    OSR_CHECK_LAUNCH("osr_roi_box_losses_fwd");
    hipLaunchKernelGGL(tr_final_reduce, dim3(1), dim3(64), 0, st, partial, RED_BLOCKS, 3, scale, 2, out3);
    OSR_CHECK_LAUNCH("osr_roi_box_losses_fwd(final)");
    return OSR_OK;
}

extern "C" osr_status osr_roi_box_losses_fwd(const float* pred_deltas, int32_t delta_stride, const float* pred_iou, int32_t iou_stride,
                                             int32_t iou_is_logit, const float* proposal_boxes, const float* gt_boxes, const int64_t* gt_classes,
                                             const float* gt_iou, int64_t m, int32_t num_classes, const float reg_weights[4], float box_weight,
                                             float iou_weight, float* out3, void* workspace, int64_t workspace_bytes, void* stream) {
    return osr_roi_box_losses_fwd_ex(pred_deltas, delta_stride, pred_iou, iou_stride, iou_is_logit, proposal_boxes, gt_boxes, gt_classes, gt_iou, m,
                                     num_classes, reg_weights, box_weight, iou_weight, nullptr, out3, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------
// PLN hinge loss forward (prototype_learning_network.py:133-187, COS distance, one prototype per class)
// ------------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void pln_loss_kernel(const float* __restrict__ emb, long long m, int d, const float* __restrict__ protos, int K, int R,
                                                       int dist_type, const long long* __restrict__ cls, const float* __restrict__ ious, float iou_thr,
                                                       float alpha, float beta, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];  // [K * R][d]: R prototypes per class, class-major
    const int KR = K * R;
    if (((KR * d) & 3) == 0 && (reinterpret_cast<uintptr_t>(protos) & 15) == 0) {  // 16-byte copies, four in flight per thread
        const int n4 = (KR * d) >> 2;
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<float4*>(s_p)[i] = reinterpret_cast<const float4*>(protos)[i];
    } else {
        for (int i = threadIdx.x; i < KR * d; i += blockDim.x) s_p[i] = protos[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float v[4] = {0.f, 0.f, 0.f, 0.f};  // intra, inter, center, rows that count (class >= 0; padding rows carry -1)
    const int W = (int)gridDim.x * nw, w = (int)blockIdx.x * nw + wid;
    for (long long k = 0; k * W < m; ++k) {
        const long long r = osr_pln_row(k, w, W);
        if (r >= m) continue;
        const long long y = cls[r];
        const float iou = ious[r];  // (both loads up front: one round trip)
        if (lane == 0 && y >= 0) v[3] += 1.f;
        if (!(y >= 0 && y < K && iou > iou_thr)) continue;  // foreground of a known class with IoU above the threshold
        const float* e = emb + r * d;
        const bool in_regs = d <= NJ * 64;  // (wave-uniform) the normalised row lives in registers for the class loop
        float eh[NJ];  // the row, read once: ||e|| from these registers, then divided in place
        float ss = 0.f;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) { const int i = lane + 64 * j; eh[j] = i < d ? e[i] : 0.f; ss += eh[j] * eh[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) eh[j] = 0.f;
            for (int i = lane; i < d; i += 64) { const float x = e[i]; ss += x * x; }
        }
        ss = osr_wave_sum(ss);
        const float den = fmaxf(sqrtf(ss), 1e-12f);
        float intra = 0.f, inter = 1000.f;  // the reference overwrites the own-class column with 1000 before the min
#pragma unroll
        for (int j = 0; j < NJ; ++j) eh[j] = eh[j] / den;
        if (in_regs) {  // the prototypes four at a time, in order; a class's distance is the minimum over its R consecutive prototypes (prototype_learning_network.py:163)
            float dist = 0.f;
            for (int k0 = 0; k0 < KR; k0 += 4) {
                float dq[4];
                osr_pln_distance_reg4<NJ>(eh, s_p + (size_t)k0 * d, d, KR - k0 < 4 ? KR - k0 : 4, lane, dist_type, dq);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kr = k0 + t;
                    if (kr < KR) {
                        const int c = kr / R, q = kr - c * R;
                        dist = (q == 0 || dq[t] < dist) ? dq[t] : dist;
                        if (q == R - 1) { if (c == (int)y) intra = dist; else inter = fminf(inter, dist); }
                    }
                }
            }
        } else {
            for (int c = 0; c < K; ++c) {
                float dist = 0.f;
                for (int q = 0; q < R; ++q) {
                    const float dq = osr_pln_distance([&](int i) { return e[i] / den; }, s_p + (size_t)(c * R + q) * d, d, lane, dist_type);
                    dist = (q == 0 || dq < dist) ? dq : dist;
                }
                if (c == (int)y) intra = dist; else inter = fminf(inter, dist);
            }
        }
        if (lane == 0) { v[0] += fmaxf(intra - alpha, 0.f); v[1] += fmaxf(beta - inter, 0.f); }
    }
    // prototype-to-prototype term, once: c_dist[k] = min over the prototypes j of OTHER classes of dist(p_k, p_j)
    // (prototype_learning_network.py:170-180: the diagonal class blocks are overwritten with 1000). Workgroup b takes prototypes b,
    // b + gridDim.x, ..., its waves every nw-th j each (all of them in workgroup 0 was a 24 us tail behind that workgroup's rows).
    __shared__ float s_cdw[16];
    for (int k = blockIdx.x; k < KR; k += gridDim.x) {  // (uniform over the workgroup)
        float cd = 1000.f;
        const float* pk = s_p + (size_t)k * d;
        for (int j = wid; j < KR; j += nw) {
            if (j / R == k / R) continue;
            cd = fminf(cd, osr_pln_distance([&](int i) { return pk[i]; }, s_p + (size_t)j * d, d, lane, dist_type));
        }
        __syncthreads();
        if (lane == 0) s_cdw[wid] = cd;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w2 = 1; w2 < nw; ++w2) cd = fminf(cd, s_cdw[w2]);
            v[2] += fmaxf(beta + alpha - cd, 0.f);
        }
    }
    tr_block_reduce_store<4>(v, partial);
}

// loss = weight / M * (sum intra + sum inter + sum center), each sum in workgroup order
__global__ __launch_bounds__(64) void pln_finish(const float* __restrict__ partial, int nblocks, float scale, float* __restrict__ out) {
    const float a = tr_wave_column_sum(partial, nblocks, 4, 0), b = tr_wave_column_sum(partial, nblocks, 4, 1);
    const float c = tr_wave_column_sum(partial, nblocks, 4, 2), rows = tr_wave_column_sum(partial, nblocks, 4, 3);
    if (threadIdx.x == 0) out[0] = ((a + b) + c) * scale / fmaxf(rows, 1.0f);
}

extern "C" osr_status osr_pln_loss_fwd_ex(const float* emb, int64_t m, int32_t d, const float* protos_normed, int32_t num_known, int32_t reps,
                                          int32_t distance_type, const int64_t* gt_classes, const float* ious, float iou_thr, float alpha, float beta,
                                          float loss_weight, float* out1, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(emb && protos_normed && gt_classes && ious && out1 && workspace, OSR_ERR_INVALID_ARG, "osr_pln_loss_fwd: null pointer");
    OSR_REQUIRE(m >= 0 && d >= 1 && num_known >= 1 && reps >= 1 && (long long)num_known * reps * d <= 36864, OSR_ERR_UNSUPPORTED,
                "osr_pln_loss_fwd: bad sizes (the prototypes must fit a 144 KB LDS table)");
    OSR_REQUIRE(distance_type >= OSR_DIST_COS && distance_type <= OSR_DIST_L2, OSR_ERR_INVALID_ARG, "osr_pln_loss_fwd: distance_type %d", distance_type);
    OSR_REQUIRE(workspace_bytes >= (int64_t)RED_BLOCKS * 4 * 4, OSR_ERR_WORKSPACE, "osr_pln_loss_fwd: workspace needs %d bytes", RED_BLOCKS * 4 * 4);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const size_t smem = (size_t)num_known * reps * d * 4;
    if (smem > 64 * 1024) {
        static osr_dev_mask attr{0};
        osr_once_per_device(attr, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pln_loss_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pln_loss_kernel<OSR_PLN_REG>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        });
    }
    if (d <= 256)
        hipLaunchKernelGGL(pln_loss_kernel<4>, dim3(RED_BLOCKS), dim3(256), smem, st, emb, (long long)m, d, protos_normed, num_known, reps, distance_type,
                           (const long long*)gt_classes, ious, iou_thr, alpha, beta, partial);
    else
        hipLaunchKernelGGL(pln_loss_kernel<OSR_PLN_REG>, dim3(RED_BLOCKS), dim3(256), smem, st, emb, (long long)m, d, protos_normed, num_known, reps, distance_type,
                           (const long long*)gt_classes, ious, iou_thr, alpha, beta, partial);
    OSR_CHECK_LAUNCH("osr_pln_loss_fwd");
    hipLaunchKernelGGL(pln_finish, dim3(1), dim3(64), 0, st, (const float*)partial, RED_BLOCKS, loss_weight, out1);
    OSR_CHECK_LAUNCH("osr_pln_loss_fwd(final)");
    return OSR_OK;
}

extern "C" osr_status osr_pln_loss_fwd(const float* emb, int64_t m, int32_t d, const float* protos_normed, int32_t num_known, const int64_t* gt_classes,
                                       const float* ious, float iou_thr, float alpha, float beta, float loss_weight, float* out1, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
    return osr_pln_loss_fwd_ex(emb, m, d, protos_normed, num_known, 1, OSR_DIST_COS, gt_classes, ious, iou_thr, alpha, beta, loss_weight, out1, workspace,
                               workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------
// softmax cross-entropy forward (softmax_classifier.py:276-285)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, long long m, int nc /* K+1 */, const long long* __restrict__ cls,
                                                      int num_classes, int K, float* __restrict__ partial) {
    float v[2] = {0.f, 0.f};  // sum of -log p[target], count
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const long long c = cls[i];
        const int t = (c >= 0 && c < K) ? (int)c : (c == num_classes ? K : -1);  // id_map: known -> itself, background -> K, others -> -1
        if (t < 0) continue;
        const float* lg = logits + i * nc;
        float mx = lg[0];
        for (int j = 1; j < nc; ++j) mx = fmaxf(mx, lg[j]);
        float s = 0.f;
        for (int j = 0; j < nc; ++j) s += expf(lg[j] - mx);
        v[0] += (logf(s) + mx) - lg[t];
        v[1] += 1.f;
    }
    tr_block_reduce_store<2>(v, partial);
}

__global__ __launch_bounds__(64) void ce_finish(const float* __restrict__ partial, int nblocks, float weight, float* __restrict__ out) {
    const float s = tr_wave_column_sum(partial, nblocks, 2, 0), c = tr_wave_column_sum(partial, nblocks, 2, 1);
    if (threadIdx.x == 0) out[0] = c > 0.f ? weight * (s / c) : 0.f;
}

extern "C" osr_status osr_softmax_ce_loss_fwd(const float* logits, int64_t m, int32_t num_known, const int64_t* gt_classes, int32_t num_classes,
                                              float loss_weight, float* out1, void* workspace, int64_t workspace_bytes, void* stream) {
    OSR_REQUIRE(logits && gt_classes && out1 && workspace, OSR_ERR_INVALID_ARG, "osr_softmax_ce_loss_fwd: null pointer");
    OSR_REQUIRE(m >= 0 && num_known >= 1 && num_known <= 1024, OSR_ERR_INVALID_ARG, "osr_softmax_ce_loss_fwd: bad sizes");
    OSR_REQUIRE(workspace_bytes >= (int64_t)RED_BLOCKS * 2 * 4, OSR_ERR_WORKSPACE, "osr_softmax_ce_loss_fwd: workspace needs %d bytes", RED_BLOCKS * 2 * 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_loss_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, logits, (long long)m, num_known + 1, (const long long*)gt_classes, num_classes, num_known,
                       (float*)workspace);
    OSR_CHECK_LAUNCH("osr_softmax_ce_loss_fwd");
    hipLaunchKernelGGL(ce_finish, dim3(1), dim3(64), 0, st, (const float*)workspace, RED_BLOCKS, loss_weight, out1);
    OSR_CHECK_LAUNCH("osr_softmax_ce_loss_fwd(final)");
    return OSR_OK;
}
