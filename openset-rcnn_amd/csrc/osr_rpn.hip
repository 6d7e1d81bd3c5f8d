// CF-RPN head tail + proposal selection for gfx950 (include/osr.h: osr_cfrpn_head_tail, osr_rpn_select).
//
// Replaces, for all images and pyramid levels at once,
//   ClsFreeRPNHead.forward after the 3x3 conv     classification_free_rpn.py:159-161
//   ClsFreeRPN._decode_proposals                   classification_free_rpn.py:591-610
//   find_top_rpn_proposals                         find_top_proposals.py:60-127
// Compile with -ffp-contract=off: the decode/clip arithmetic must round exactly like the oracle's.
#include "osr_common.h"

// ------------------------------------------------------------------------------------------------------
// head tail: one wave per pixel row of t (c channels, channels-last). 5 dot products + sum of squares.
// ------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void cfrpn_tail_kernel(const T* __restrict__ t, long long rows, int c,
                                                         const float* __restrict__ w_delta, const float* __restrict__ b_delta,
                                                         const float* __restrict__ w_ctr, const float* __restrict__ b_ctr,
                                                         float* __restrict__ deltas, float* __restrict__ ctr) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [5][c]
    for (int i = threadIdx.x; i < 5 * c; i += blockDim.x) s_w[i] = i < 4 * c ? w_delta[i] : w_ctr[i - 4 * c];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (long long r = (long long)blockIdx.x * nw + wid; r < rows; r += (long long)gridDim.x * nw) {
        float ss = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f;
        const T* row = t + r * c;
        for (int k = lane * 4; k < c; k += 256) {
            float v[4];
            if constexpr (sizeof(T) == 4) {
                float4 q = *reinterpret_cast<const float4*>(row + k);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
                typedef T t4 __attribute__((ext_vector_type(4)));
                t4 q = *reinterpret_cast<const t4*>(row + k);
                v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ss += v[j] * v[j];
                d0 += v[j] * s_w[k + j];
                d1 += v[j] * s_w[c + k + j];
                d2 += v[j] * s_w[2 * c + k + j];
                d3 += v[j] * s_w[3 * c + k + j];
                d4 += v[j] * s_w[4 * c + k + j];
            }
        }
        ss = osr_wave_sum(ss); d0 = osr_wave_sum(d0); d1 = osr_wave_sum(d1);
        d2 = osr_wave_sum(d2); d3 = osr_wave_sum(d3); d4 = osr_wave_sum(d4);
        if (lane == 0) {
            // F.normalize: t / max(||t||_2, 1e-12); the 1x1 convs are linear, so scale the dot products
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            float4 o = make_float4(d0 * inv + b_delta[0], d1 * inv + b_delta[1], d2 * inv + b_delta[2], d3 * inv + b_delta[3]);
            *reinterpret_cast<float4*>(deltas + r * 4) = o;
            const float z = d4 * inv + b_ctr[0];
            ctr[r] = 1.0f / (1.0f + expf(-z));
        }
    }
}

extern "C" osr_status osr_cfrpn_head_tail(const void* t, int32_t t_dtype, int64_t rows, int32_t c, const float* w_delta,
                                          const float* b_delta, const float* w_ctr, const float* b_ctr, float* deltas,
                                          float* ctr, void* stream) {
    OSR_REQUIRE(t && w_delta && b_delta && w_ctr && b_ctr && deltas && ctr, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_tail: null pointer");
    OSR_REQUIRE(osr_dtype_ok(t_dtype), OSR_ERR_INVALID_ARG, "osr_cfrpn_head_tail: bad dtype %d", t_dtype);
    OSR_REQUIRE(c > 0 && c % 4 == 0 && c <= 2048, OSR_ERR_UNSUPPORTED, "osr_cfrpn_head_tail: c must be a multiple of 4 and <= 2048, got %d", c);
    OSR_REQUIRE(rows >= 0, OSR_ERR_INVALID_ARG, "osr_cfrpn_head_tail: rows < 0");
    if (rows == 0) return OSR_OK;
    long long blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    size_t smem = (size_t)5 * c * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (t_dtype == OSR_F32)
        hipLaunchKernelGGL(cfrpn_tail_kernel<float>, dim3((unsigned)blocks), dim3(256), smem, st, (const float*)t, rows, c, w_delta, b_delta, w_ctr, b_ctr, deltas, ctr);
    else if (t_dtype == OSR_F16)
        hipLaunchKernelGGL(cfrpn_tail_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), smem, st, (const f16_t*)t, rows, c, w_delta, b_delta, w_ctr, b_ctr, deltas, ctr);
    else
        hipLaunchKernelGGL(cfrpn_tail_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), smem, st, (const bf16_t*)t, rows, c, w_delta, b_delta, w_ctr, b_ctr, deltas, ctr);
    OSR_CHECK_LAUNCH("osr_cfrpn_head_tail");
    return OSR_OK;
}

// ------------------------------------------------------------------------------------------------------
// proposal selection
// ------------------------------------------------------------------------------------------------------
#define SEL_THREADS 1024
#define SEL_MAXK 2048
#define SEL_BINS 2048  // histogram bins of a radix-select pass (11 key bits)

struct SelLevels {
    int num_levels, num_anchors;
    int h[OSR_MAX_LEVELS], w[OSR_MAX_LEVELS], stride[OSR_MAX_LEVELS];
    long long offset[OSR_MAX_LEVELS];
    int klevel[OSR_MAX_LEVELS];  // min(h*w*a, topk)
    int koff[OSR_MAX_LEVELS];    // prefix of klevel
    int aoff[OSR_MAX_LEVELS];    // prefix of h*w*a (index into the image's concatenated anchor list)
    int cap;
};

// descending bitonic sort of 64-bit keys in LDS, n power of two
__device__ __forceinline__ void bitonic_desc(unsigned long long* buf, int n) {
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

// hist[digit] += 1 for the lanes with `on`, called by all 64 lanes of a wave. Two rounds of "the first pending lane's digit: every
// lane that shares it is added by ONE atomic", then one atomic per lane that is still pending. Same histogram as 64 plain atomics.
__device__ __forceinline__ void sel_hist_add(int* hist, int digit, bool on) {
    const int lane = threadIdx.x & 63;
    unsigned long long act = __ballot(on);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (!act) return;  // wave-uniform
        const int leader = __ffsll((long long)act) - 1;
        const int d0 = __shfl(digit, leader, 64);
        const bool mine = on && digit == d0;
        const unsigned long long m = __ballot(mine);
        if (lane == leader) atomicAdd(&hist[d0], __popcll(m));
        if (mine) on = false;
        act &= ~m;
    }
    if (on) atomicAdd(&hist[digit], 1);
}

// Stage A: grid (level, image). Stable top-k by radix select + bitonic sort, ltrb decode, filters.
// Writes staging arrays st_box/st_score/st_src/st_flag at [img][koff[l] + rank].
__global__ __launch_bounds__(SEL_THREADS) void rpn_select_kernel(SelLevels lv, const float* __restrict__ cell_anchors,
                                                                 const float* __restrict__ ctr, const float* __restrict__ deltas,
                                                                 int n_img, const int* __restrict__ image_hw, float min_box_size,
                                                                 int decode_mode, float4 rw, float* __restrict__ st_box,
                                                                 float* __restrict__ st_score, int* __restrict__ st_src, int* __restrict__ st_flag,
                                                                 int* __restrict__ status_flags) {
    const int l = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int A = lv.num_anchors, W = lv.w[l];
    const int cnt = lv.h[l] * W * A;
    const int k = lv.klevel[l];
    const float* sc = ctr + lv.offset[l] + (long long)img * cnt;
    const float* dl = deltas + (lv.offset[l] + (long long)img * cnt) * 4;

    __shared__ unsigned long long s_sel[SEL_MAXK];
    __shared__ __attribute__((aligned(16))) int s_hist[SEL_BINS];
    __shared__ int s_scan[32];
    __shared__ unsigned int s_prefix, s_remaining;

    // Four scores per thread and load (16 bytes per lane) where the segment allows it: the level's score run must be 16-byte
    // aligned and a multiple of 4 long (true for p2..p4 of every image; the small levels take the scalar form).
    const bool vec4 = (cnt & 3) == 0 && (((uintptr_t)sc) & 15) == 0;
    const int per = vec4 ? 4 : 1;
    const int nit = (cnt + SEL_THREADS * per - 1) / (SEL_THREADS * per);  // uniform trip count of the strided loops below

    // ---- 1. radix select: key T of the k-th largest element (3 passes: 11 + 11 + 10 key bits, MSB first). The scores of a level are sigmoid
    //      outputs: their leading key byte is the same for almost every anchor, so one LDS atomic per element would put all 67 200
    //      adds of a p2 pass on one histogram bin, one behind the other. sel_hist_add adds a wave's most common digits once per wave. ----
    unsigned int prefix = 0, mask = 0;
    int remaining = k;  // rank (1-based, from the top) still to resolve inside the current prefix bucket
    if (cnt > k) {
        // three passes over the scores: 11 + 11 + 10 key bits, MSB first (four 8-bit passes read the 269 KB of a p2 level once more)
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0, nbins = pass == 2 ? 1024 : 2048;
            for (int i = tid; i < SEL_BINS; i += blockDim.x) s_hist[i] = 0;
            __syncthreads();
            for (int it = 0; it < nit; ++it) {
                const int i = (it * SEL_THREADS + tid) * per;
                unsigned int key[4] = {0u, 0u, 0u, 0u};
                if (i < cnt) {
                    if (vec4) {
                        const float4 v = *reinterpret_cast<const float4*>(sc + i);
                        key[0] = osr_float_key(v.x); key[1] = osr_float_key(v.y); key[2] = osr_float_key(v.z); key[3] = osr_float_key(v.w);
                    } else key[0] = osr_float_key(sc[i]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < per) sel_hist_add(s_hist, (int)((key[e] >> shift) & (unsigned)(nbins - 1)), i < cnt && (key[e] & mask) == prefix);
            }
            __syncthreads();
            if (tid < 64) {
                // the digit d whose bucket holds the rank: the largest d with sum(hist[d..]) >= remaining (d = 0 when only the whole
                // histogram reaches it). One wave: SEL_BINS / 64 bins per lane, suffix sums over the lanes, then the lane that holds
                // the crossing walks its own bins (a serial walk over all bins: dependent LDS reads, 7 us per 256 of them).
                constexpr int BPL = SEL_BINS / 64;
                int sum = 0;
#pragma unroll
                for (int j = 0; j < BPL; j += 4) {
                    const int4 h4 = *reinterpret_cast<const int4*>(&s_hist[tid * BPL + j]);
                    sum += h4.x + h4.y + h4.z + h4.w;
                }
                int inc = sum;  // becomes the sum over this lane's bins and all higher lanes'
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const int t = __shfl_down(inc, dd, 64);
                    if (tid + dd < 64) inc += t;
                }
                const int above = inc - sum;
                const bool here = above < remaining && remaining <= inc;
                if (here) {
                    int acc = above, d = tid * BPL;
                    for (int j = BPL - 1; j >= 0; --j) {
                        const int hj = s_hist[tid * BPL + j];
                        if (acc + hj >= remaining) { d = tid * BPL + j; break; }
                        acc += hj;
                    }
                    s_prefix = prefix | ((unsigned int)d << shift);
                    s_remaining = remaining - acc;
                }
                if (!__ballot(here) && tid == 0) {  // (cannot happen while the bucket holds >= remaining elements: the serial walk's d = 0 exit)
                    s_prefix = prefix;
                    s_remaining = remaining - (inc - s_hist[0]);
                }
            }
            __syncthreads();
            prefix = s_prefix;
            remaining = s_remaining;
            mask |= (unsigned)(nbins - 1) << shift;
            __syncthreads();
        }
    }
    // Now: elements with key > prefix are all selected; of those with key == prefix the first `remaining`
    // in index order are selected (stable tie rule). If cnt <= k everything is selected.
    const unsigned int T = prefix;
    const bool all = cnt <= k;

    // ---- 2. ordered compaction into s_sel as (key << 32) | ~index: strictly greater keys to [0, ngt), ties to [ngt, k). The k-th
    //      largest key is T and `remaining` of its ties are needed, so exactly k - remaining keys are greater: no counting pass ----
    int base_gt = 0, base_eq = 0;  // running counts (uniform across the block)
    const int ngt_total = all ? 0 : k - remaining;
    for (int it = 0; it < nit; ++it) {
        const int i = (it * SEL_THREADS + tid) * per;
        unsigned int key[4] = {0u, 0u, 0u, 0u};
        int gt = 0, eq = 0;  // how many of this thread's (up to four, consecutive) elements are greater / tie
        if (i < cnt) {
            if (vec4) {
                const float4 v = *reinterpret_cast<const float4*>(sc + i);
                key[0] = osr_float_key(v.x); key[1] = osr_float_key(v.y); key[2] = osr_float_key(v.z); key[3] = osr_float_key(v.w);
            } else key[0] = osr_float_key(sc[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < per) { if (all) gt += 1; else { gt += key[e] > T; eq += key[e] == T; } }
        }
        int tot;
        const int packed = osr_block_excl_scan(gt | (eq << 16), s_scan, &tot);
        int pg = base_gt + (packed & 0xffff), pe = base_eq + (packed >> 16);
        if (i < cnt) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < per) {
                    const unsigned long long comp = ((unsigned long long)key[e] << 32) | (unsigned int)(0xffffffffu - (unsigned int)(i + e));
                    const bool g = all || key[e] > T, q = !all && key[e] == T;
                    if (g) { if (pg < SEL_MAXK) s_sel[pg] = comp; ++pg; }
                    if (q) { if (pe < remaining && ngt_total + pe < SEL_MAXK) s_sel[ngt_total + pe] = comp; ++pe; }
                }
        }
        base_gt += tot & 0xffff;
        base_eq += tot >> 16;
        if (!all && base_eq >= remaining && base_gt >= ngt_total) break;  // uniform
    }
    int kp = 1;
    while (kp < k) kp <<= 1;
    for (int i = k + tid; i < kp; i += blockDim.x) s_sel[i] = 0ull;  // padding sorts last
    __syncthreads();

    // ---- 3. sort the survivors: score descending, index ascending ----
    bitonic_desc(s_sel, kp);

    // ---- 4. decode + filters ----
    const float ih = (float)image_hw[img * 2 + 0], iw = (float)image_hw[img * 2 + 1];
    const float fstride = (float)lv.stride[l];
    bool bad = false;
    for (int j = tid; j < k; j += blockDim.x) {
        const unsigned long long comp = s_sel[j];
        const int idx = (int)(0xffffffffu - (unsigned int)(comp & 0xffffffffull));
        const float s = sc[idx];
        const float4 d = *reinterpret_cast<const float4*>(dl + (long long)idx * 4);
        const int a = idx % A, cell = idx / A;
        const float sx = (float)(cell % W) * fstride, sy = (float)(cell / W) * fstride;
        const float* ca = cell_anchors + ((long long)l * A + a) * 4;
        const float ax1 = sx + ca[0], ay1 = sy + ca[1], ax2 = sx + ca[2], ay2 = sy + ca[3];
        const float aw = ax2 - ax1, ah = ay2 - ay1;
        float x1, y1, x2, y2;
        if (decode_mode == 0) {
            // [d2] Box2BoxTransformLinear(normalize_by_size=True).apply_deltas (the CF-RPN, classification_free_rpn.py:607)
            const float cx = 0.5f * (ax1 + ax2), cy = 0.5f * (ay1 + ay2);
            const float dl_ = fmaxf(d.x, 0.f) * aw, dt_ = fmaxf(d.y, 0.f) * ah, dr_ = fmaxf(d.z, 0.f) * aw, db_ = fmaxf(d.w, 0.f) * ah;
            x1 = cx - dl_; y1 = cy - dt_; x2 = cx + dr_; y2 = cy + db_;
        } else {
            // [d2] Box2BoxTransform(weights).apply_deltas (the stock RPN of Base-RCNN-FPN.yaml: weights 1,1,1,1): dw, dh clamped
            // from above at log(1000/16)
            const float cx = ax1 + 0.5f * aw, cy = ay1 + 0.5f * ah;
            const float kClamp = 4.135166556742356f;
            const float dx = d.x / rw.x, dy = d.y / rw.y, dw = fminf(d.z / rw.z, kClamp), dh = fminf(d.w / rw.w, kClamp);
            const float pcx = dx * aw + cx, pcy = dy * ah + cy, pw = expf(dw) * aw, ph = expf(dh) * ah;
            x1 = pcx - 0.5f * pw; y1 = pcy - 0.5f * ph; x2 = pcx + 0.5f * pw; y2 = pcy + 0.5f * ph;
        }
        // relu(NaN) / clamp(NaN) must stay NaN like torch's: fmaxf / fminf drop NaN, so test the raw deltas too
        const bool dnan = (d.x != d.x) || (d.y != d.y) || (d.z != d.z) || (d.w != d.w);
        const bool valid = osr_finite(x1) && osr_finite(y1) && osr_finite(x2) && osr_finite(y2) && osr_finite(s) && !dnan;
        bad |= !valid;
        x1 = fminf(fmaxf(x1, 0.f), iw); y1 = fminf(fmaxf(y1, 0.f), ih);
        x2 = fminf(fmaxf(x2, 0.f), iw); y2 = fminf(fmaxf(y2, 0.f), ih);
        const bool keep = valid && (x2 - x1 > min_box_size) && (y2 - y1 > min_box_size);
        const long long o = (long long)img * lv.cap + lv.koff[l] + j;
        *reinterpret_cast<float4*>(st_box + o * 4) = make_float4(x1, y1, x2, y2);
        st_score[o] = s;
        st_src[o] = lv.aoff[l] + idx;
        st_flag[o] = keep ? 1 : 0;
    }
    if (bad) atomicOr(status_flags, 1);
}

// Stage B: one block per image; order-preserving compaction of the staged slots.
__global__ __launch_bounds__(SEL_THREADS) void rpn_compact_kernel(int cap, const float* __restrict__ st_box,
                                                                  const float* __restrict__ st_score, const int* __restrict__ st_src,
                                                                  const int* __restrict__ st_flag, float* __restrict__ boxes,
                                                                  float* __restrict__ scores, int* __restrict__ src_index,
                                                                  int* __restrict__ batch_idx, int* __restrict__ counts, SelLevels lv,
                                                                  int* __restrict__ level_out) {
    const int img = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_scan[32];
    const long long base = (long long)img * cap;
    int running = 0;
    for (int i0 = 0; i0 < cap; i0 += blockDim.x) {
        const int i = i0 + tid;
        const int f = i < cap ? st_flag[base + i] : 0;
        int tot;
        const int pos = running + osr_block_excl_scan(f, s_scan, &tot);
        if (f) {
            *reinterpret_cast<float4*>(boxes + (base + pos) * 4) = *reinterpret_cast<const float4*>(st_box + (base + i) * 4);
            scores[base + pos] = st_score[base + i];
            src_index[base + pos] = st_src[base + i];
            batch_idx[base + pos] = img;
            if (level_out) {  // slot i of the staging array belongs to the level whose [koff, koff + klevel) holds it
                int l = 0;
                while (l + 1 < lv.num_levels && i >= lv.koff[l + 1]) ++l;
                level_out[base + pos] = l;
            }
        }
        running += tot;
    }
    for (int i = running + tid; i < cap; i += blockDim.x) {
        *reinterpret_cast<float4*>(boxes + (base + i) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        scores[base + i] = 0.f;
        src_index[base + i] = -1;
        batch_idx[base + i] = -1;
        if (level_out) level_out[base + i] = -1;
    }
    if (tid == 0) counts[img] = running;
}

static bool fill_levels(const osr_rpn_levels* in, int topk, SelLevels* o) {
    if (!in || in->num_levels < 1 || in->num_levels > OSR_MAX_LEVELS || in->num_anchors < 1 || topk < 1 || topk > SEL_MAXK) return false;
    o->num_levels = in->num_levels;
    o->num_anchors = in->num_anchors;
    int ko = 0;
    long long ao = 0;
    for (int l = 0; l < in->num_levels; ++l) {
        if (in->h[l] < 1 || in->w[l] < 1 || in->stride[l] < 1 || in->offset[l] < 0) return false;
        long long cnt = (long long)in->h[l] * in->w[l] * in->num_anchors;
        if (cnt > (1ll << 30) || ao + cnt > (1ll << 30)) return false;
        o->h[l] = in->h[l]; o->w[l] = in->w[l]; o->stride[l] = in->stride[l]; o->offset[l] = in->offset[l];
        o->klevel[l] = (int)(cnt < topk ? cnt : topk);
        o->koff[l] = ko;
        o->aoff[l] = (int)ao;
        ko += o->klevel[l];
        ao += cnt;
    }
    o->cap = ko;
    return true;
}

extern "C" int32_t osr_rpn_select_capacity(const osr_rpn_levels* lv, int32_t pre_nms_topk) {
    SelLevels s;
    if (!fill_levels(lv, pre_nms_topk, &s)) { osr_set_error("osr_rpn_select_capacity: bad level table / topk (1..%d)", SEL_MAXK); return OSR_ERR_INVALID_ARG; }
    return s.cap;
}

extern "C" int64_t osr_rpn_select_workspace_bytes(const osr_rpn_levels* lv, int32_t n, int32_t pre_nms_topk) {
    SelLevels s;
    if (!fill_levels(lv, pre_nms_topk, &s) || n < 1) { osr_set_error("osr_rpn_select_workspace_bytes: bad arguments"); return OSR_ERR_INVALID_ARG; }
    return (int64_t)n * s.cap * (4 * 4 + 4 + 4 + 4);
}

extern "C" osr_status osr_rpn_select_ex(const osr_rpn_levels* lv, const float* cell_anchors, const float* ctr, const float* deltas,
                                        int32_t n, const int32_t* image_hw, int32_t pre_nms_topk, float min_box_size, int32_t decode_mode,
                                        const float reg_weights[4], float* boxes, float* scores, int32_t* src_index, int32_t* batch_idx,
                                        int32_t* level_out, int32_t* counts, int32_t* status_flags, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
    SelLevels s;
    OSR_REQUIRE(fill_levels(lv, pre_nms_topk, &s), OSR_ERR_INVALID_ARG, "osr_rpn_select: bad level table / topk (1..%d)", SEL_MAXK);
    OSR_REQUIRE(cell_anchors && ctr && deltas && image_hw && boxes && scores && src_index && batch_idx && counts && status_flags && workspace,
                OSR_ERR_INVALID_ARG, "osr_rpn_select: null pointer");
    OSR_REQUIRE(n >= 1 && n <= 65535, OSR_ERR_INVALID_ARG, "osr_rpn_select: n out of range");
    OSR_REQUIRE(decode_mode == 0 || (decode_mode == 1 && reg_weights && reg_weights[0] > 0.f && reg_weights[1] > 0.f && reg_weights[2] > 0.f && reg_weights[3] > 0.f),
                OSR_ERR_INVALID_ARG, "osr_rpn_select: decode_mode 0 (ltrb) or 1 (Box2BoxTransform with positive weights)");
    const int64_t need = (int64_t)n * s.cap * 28;
    OSR_REQUIRE(workspace_bytes >= need, OSR_ERR_WORKSPACE, "osr_rpn_select: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    OSR_REQUIRE(((uintptr_t)workspace & 15) == 0, OSR_ERR_INVALID_ARG, "osr_rpn_select: workspace must be 16-byte aligned");
    char* ws = (char*)workspace;
    float* st_box = (float*)ws;
    float* st_score = (float*)(ws + (int64_t)n * s.cap * 16);
    int* st_src = (int*)(ws + (int64_t)n * s.cap * 20);
    int* st_flag = (int*)(ws + (int64_t)n * s.cap * 24);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(status_flags, 0, sizeof(int32_t), st) != hipSuccess) { osr_set_error("osr_rpn_select: memset failed"); return OSR_ERR_LAUNCH; }
    const float4 rw = decode_mode == 1 ? make_float4(reg_weights[0], reg_weights[1], reg_weights[2], reg_weights[3]) : make_float4(1.f, 1.f, 1.f, 1.f);
    hipLaunchKernelGGL(rpn_select_kernel, dim3(s.num_levels, n), dim3(SEL_THREADS), 0, st, s, cell_anchors, ctr, deltas, n, image_hw,
                       min_box_size, decode_mode, rw, st_box, st_score, st_src, st_flag, status_flags);
    OSR_CHECK_LAUNCH("osr_rpn_select(select)");
    hipLaunchKernelGGL(rpn_compact_kernel, dim3(n), dim3(SEL_THREADS), 0, st, s.cap, st_box, st_score, st_src, st_flag, boxes, scores,
                       src_index, batch_idx, counts, s, level_out);
    OSR_CHECK_LAUNCH("osr_rpn_select(compact)");
    return OSR_OK;
}

extern "C" osr_status osr_rpn_select(const osr_rpn_levels* lv, const float* cell_anchors, const float* ctr, const float* deltas,
                                     int32_t n, const int32_t* image_hw, int32_t pre_nms_topk, float min_box_size, float* boxes,
                                     float* scores, int32_t* src_index, int32_t* batch_idx, int32_t* counts, int32_t* status_flags,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    return osr_rpn_select_ex(lv, cell_anchors, ctr, deltas, n, image_hw, pre_nms_topk, min_box_size, 0, nullptr, boxes, scores, src_index, batch_idx,
                             nullptr, counts, status_flags, workspace, workspace_bytes, stream);
}
