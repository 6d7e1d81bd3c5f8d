/*
 * C restatement of the loop-heavy hot-path ops (RoIAlign, NMS, top-k) -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker. PARITY UNPINNED: the reference ships no golden vectors and cannot be imported here
 * (SURVEY.md 8c); the algorithms below restate torchvision 0.11's CPU kernels ([d2-mem], pinned by
 * /root/reference/README.md:20-33) at the reference's call sites and are cross-checked against the
 * independent numpy restatement in osr_oracle.py and against analytic known answers.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off; no fast-math, so fp32 results are
 * reproducible and comparable bit-for-bit with the HIP kernels' index outputs).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- RoIAlign forward, torchvision semantics (call site: osrcnn_roi_heads.py:306 via d2 ROIPooler) ----
 * feat: (N,C,H,W) fp32 NCHW. rois: (K,5) [batch,x1,y1,x2,y2]. out: (K,C,P,P).
 * aligned=1: coordinates*scale-0.5, no min-size clamp; sampling_ratio<=0 => ceil(roi/P) samples per bin;
 * samples with y<-1||y>H||x<-1||x>W contribute 0; count=max(gh*gw,1). */
void osr_oracle_roi_align(const float* feat, int N, int C, int H, int W, const float* rois, int K, float scale,
                          int P, int sampling_ratio, int aligned, float* out) {
    (void)N;
#pragma omp parallel for schedule(dynamic, 8)
    for (int r = 0; r < K; ++r) {
        const float* roi = rois + (size_t)r * 5;
        int b = (int)roi[0];
        float off = aligned ? 0.5f : 0.0f;
        float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
        float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
        float rw = ew - sw, rh = eh - sh;
        if (!aligned) {
            rw = rw > 1.f ? rw : 1.f;
            rh = rh > 1.f ? rh : 1.f;
        }
        float bh = rh / (float)P, bw = rw / (float)P;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
        float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);
        for (int c = 0; c < C; ++c) {
            const float* f = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + ph * bh + ((float)iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + pw * bw + ((float)ix + .5f) * bw / (float)gw;
                            if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                            float yy = y <= 0 ? 0 : y, xx = x <= 0 ? 0 : x;
                            int yl = (int)yy, xl = (int)xx, yh, xh;
                            if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                            if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
                            float ly = yy - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
                            float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                            acc += w1 * f[yl * W + xl] + w2 * f[yl * W + xh] + w3 * f[yh * W + xl] + w4 * f[yh * W + xh];
                        }
                    }
                    out[(((size_t)r * C + c) * P + ph) * P + pw] = acc / count;
                }
        }
    }
}

/* ---- stable descending argsort of fp32 keys (ties: lower index first; -0 == +0) ---- */
typedef struct { float v; int64_t i; } kv_t;
static int kv_cmp(const void* a, const void* b) {
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}
void osr_oracle_argsort_desc(const float* v, int64_t n, int64_t* order) {
    kv_t* t = (kv_t*)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { t[i].v = v[i]; t[i].i = i; }
    qsort(t, (size_t)n, sizeof(kv_t), kv_cmp);
    for (int64_t i = 0; i < n; ++i) order[i] = t[i].i;
    free(t);
}

/* ---- greedy NMS, torchvision CPU kernel semantics (call sites: softmax_classifier.py:93,154;
 * osrcnn_fast_rcnn.py:135). Returns number kept; keep[] holds indices in descending-score order. ---- */
int64_t osr_oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep) {
    if (n <= 0) return 0;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    osr_oracle_argsort_desc(scores, n, order);
    for (int64_t i = 0; i < n; ++i) area[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3], ia = area[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1 > 0.f ? xx2 - xx1 : 0.f;
            float h = yy2 - yy1 > 0.f ? yy2 - yy1 : 0.f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    free(order); free(sup); free(area);
    return nk;
}

/* ---- batched (per-class) NMS, vanilla semantics; keep[] in global descending-score order ---- */
int64_t osr_oracle_batched_nms(const float* boxes, const float* scores, const int64_t* cls, int64_t n, float thr,
                               int64_t* keep) {
    if (n <= 0) return 0;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
    osr_oracle_argsort_desc(scores, n, order);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float ia = (ix2 - ix1) * (iy2 - iy1);
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j] || cls[j] != cls[i]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1 > 0.f ? xx2 - xx1 : 0.f;
            float h = yy2 - yy1 > 0.f ? yy2 - yy1 : 0.f;
            float inter = w * h;
            float aj = (boxes[4 * j + 2] - boxes[4 * j]) * (boxes[4 * j + 3] - boxes[4 * j + 1]);
            float ovr = inter / (ia + aj - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    free(order); free(sup);
    return nk;
}
