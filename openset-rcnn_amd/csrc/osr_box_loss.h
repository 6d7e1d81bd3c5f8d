// Box regression losses of the training step, value and gradient, shared by the CF-RPN and the RoI box head kernels.
//
// The reference selects them by name in _dense_box_regression_loss_w_iou (/root/reference/openset_rcnn/modeling/
// box_regression_w_iou.py:13-85): "smooth_l1" acts on the deltas; "iou" is 1 - clamp(IoU, 1e-6) of the decoded box (:49-61);
// "giou" / "diou" / "ciou" call fvcore.nn.giou_loss and detectron2.layers.diou_loss / ciou_loss (third-party, not vendored in
// /root/reference; their published definitions, eps = 1e-7, are restated here: Rezatofighi et al. 2019, Zheng et al. 2020).
// Both shipped yaml files use "iou" for the CF-RPN and "smooth_l1" with beta 0 everywhere else.
#pragma once
#include <hip/hip_runtime.h>

enum OsrBoxLoss { OSR_LOSS_IOU = 0, OSR_LOSS_SMOOTH_L1 = 1, OSR_LOSS_GIOU = 2, OSR_LOSS_DIOU = 3, OSR_LOSS_CIOU = 4 };

// fvcore smooth_l1_loss: beta < 1e-5 -> |x|; else 0.5 x^2 / beta below beta, |x| - 0.5 beta above.
__device__ __forceinline__ float osr_smooth_l1(float x, float beta) {
    const float ax = fabsf(x);
    if (beta < 1e-5f) return ax;
    return ax < beta ? 0.5f * x * x / beta : ax - 0.5f * beta;
}
__device__ __forceinline__ float osr_smooth_l1_grad(float x, float beta) {
    if (beta >= 1e-5f && fabsf(x) < beta) return x / beta;
    return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
}

// Loss of a predicted box p = (x1, y1, x2, y2) against the target g, and its gradient w.r.t. the four coordinates of p.
// type: OSR_LOSS_IOU / GIOU / DIOU / CIOU. The enclosing-box / intersection corners pick the predicted coordinate on strict
// inequality (a tie is a set of measure zero for float boxes; torch's min / max share or route the gradient there).
template <bool GRAD>
__device__ __forceinline__ float osr_box_loss(int type, float4 p, float4 g, float (&dp)[4]) {
    const float eps = 1e-7f;
    const float iw = fminf(p.z, g.z) - fmaxf(p.x, g.x), ih = fminf(p.w, g.w) - fmaxf(p.y, g.y);
    const bool hit = iw > 0.f && ih > 0.f;
    const float I = hit ? iw * ih : 0.f;
    const float pw = p.z - p.x, ph = p.w - p.y, gw = g.z - g.x, gh = g.w - g.y;
    const float A = pw * ph, U = A + gw * gh - I;
    // d I, d A w.r.t. (x1, y1, x2, y2)
    float dI[4] = {0.f, 0.f, 0.f, 0.f};
    if (GRAD && hit) { dI[0] = p.x > g.x ? -ih : 0.f; dI[1] = p.y > g.y ? -iw : 0.f; dI[2] = p.z < g.z ? ih : 0.f; dI[3] = p.w < g.w ? iw : 0.f; }
    const float dA[4] = {-ph, -pw, ph, pw};
    if (GRAD) { dp[0] = dp[1] = dp[2] = dp[3] = 0.f; }
    if (type == OSR_LOSS_IOU) {
        // [d2] pairwise_iou: inter / union where inter > 0, else 0; clamp(min = 1e-6): constant (zero gradient) below the clamp
        const float iou = hit ? I / U : 0.f;
        if (GRAD && iou > 1e-6f) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dp[q] = -((dI[q] * (U + I) - I * dA[q]) / (U * U));  // d(I/U) = (dI U - I (dA - dI)) / U^2
        }
        return 1.0f - fmaxf(iou, 1e-6f);
    }
    const float Ue = U + eps, iou = I / Ue;
    float diou[4];
    if (GRAD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) diou[q] = (dI[q] * Ue - I * (dA[q] - dI[q])) / (Ue * Ue);
    }
    const float cx1 = fminf(p.x, g.x), cy1 = fminf(p.y, g.y), cx2 = fmaxf(p.z, g.z), cy2 = fmaxf(p.w, g.w);
    const float cw = cx2 - cx1, ch = cy2 - cy1;
    // d cw, d ch w.r.t. the predicted coordinates
    const float dcw[4] = {p.x < g.x ? -1.f : 0.f, 0.f, p.z > g.z ? 1.f : 0.f, 0.f};
    const float dch[4] = {0.f, p.y < g.y ? -1.f : 0.f, 0.f, p.w > g.w ? 1.f : 0.f};
    if (type == OSR_LOSS_GIOU) {
        const float C = cw * ch, Ce = C + eps;
        if (GRAD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dC = dcw[q] * ch + cw * dch[q], dU = dA[q] - dI[q];
                dp[q] = -diou[q] + ((dC - dU) * Ce - (C - U) * dC) / (Ce * Ce);
            }
        }
        return 1.0f - (iou - (C - U) / Ce);
    }
    // DIoU / CIoU: squared centre distance over the squared diagonal of the enclosing box
    const float L = cw * cw + ch * ch + eps;
    const float ex = 0.5f * (p.x + p.z) - 0.5f * (g.x + g.z), ey = 0.5f * (p.y + p.w) - 0.5f * (g.y + g.w);
    const float D = ex * ex + ey * ey;
    float loss = 1.0f - iou + D / L;
    float v = 0.f, alpha = 0.f, dat = 0.f;
    if (type == OSR_LOSS_CIOU) {
        const float k = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
        dat = atanf(gw / gh) - atanf(pw / ph);
        v = k * dat * dat;
        alpha = v / (1.0f - iou + v + eps);  // (computed under no_grad in the reference implementation: a constant for the gradient)
        loss += alpha * v;
    }
    if (GRAD) {
        const float dD[4] = {ex, ey, ex, ey};  // d D / d x1 = 2 ex * 0.5
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dL = 2.0f * cw * dcw[q] + 2.0f * ch * dch[q];
            dp[q] = -diou[q] + (dD[q] * L - D * dL) / (L * L);
        }
        if (type == OSR_LOSS_CIOU) {
            const float k = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
            const float den = pw * pw + ph * ph;
            // d atan(pw / ph): / d pw = ph / den, / d ph = -pw / den ; v = k (at_g - at_p)^2
            const float dv_dpw = -2.0f * k * dat * (ph / den), dv_dph = 2.0f * k * dat * (pw / den);
            dp[0] += alpha * (-dv_dpw); dp[2] += alpha * dv_dpw;
            dp[1] += alpha * (-dv_dph); dp[3] += alpha * dv_dph;
        }
    }
    return loss;
}
