/*
 * osr.h -- C ABI of libosr_hip.so: the MI355X (gfx950) implementation of Openset R-CNN's per-image
 * detection hot path.
 *
 * The reference (Yifei-Y/Openset-RCNN) has NO FFI of its own: its boundary is detectron2's registries and
 * nn.Module call signatures. Each entry point below therefore cites the reference call site
 * (file:line under /root/reference) -- or the un-vendored detectron2/torchvision primitive invoked there,
 * marked [d2] -- whose arithmetic it replaces. INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns every buffer including workspace (sizes from osr_*_workspace_bytes); the library never
 *     allocates, frees or synchronises; all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 = OSR_OK, negative = error (osr_last_error() gives a thread-local message);
 *     no C++ exception crosses the boundary; no global mutable state (re-entrant across threads/streams);
 *   - activations are NHWC ("channels last"); element types are named by osr_dtype.
 */
#ifndef OSR_H_
#define OSR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSR_ABI_VERSION 1

typedef int32_t osr_status;
enum {
    OSR_OK = 0,
    OSR_ERR_INVALID_ARG = -1,
    OSR_ERR_UNSUPPORTED = -2,
    OSR_ERR_LAUNCH = -3,
    OSR_ERR_WORKSPACE = -4
};

typedef enum { OSR_F32 = 0, OSR_F16 = 1, OSR_BF16 = 2 } osr_dtype;

#define OSR_MAX_LEVELS 8

int32_t osr_abi_version(void);
const char* osr_last_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * Pre-processing  ([d2] GeneralizedRCNN.preprocess_image + ImageList.from_tensors, called from train.py:135)
 * src: (n,3,h,w) NCHW, uint8 (src_is_u8=1) or float32. dst: (n, hp+6, wpad, 4) NHWC fp16/bf16 where the
 * normalised image sits at row offset 3 / column offset 3, everything else (3-pixel halo, the /32 padding,
 * the 4th channel) is zero. wpad = osr_stem_padded_width(wp). This is the layout the 7x7/s2 stem reads
 * without bounds checks.
 * --------------------------------------------------------------------------------------------------------- */
int32_t osr_stem_padded_width(int32_t wp);
osr_status osr_preprocess(const void* src, int32_t src_is_u8, int32_t n, int32_t h, int32_t w, int32_t hp, int32_t wp,
                          const float mean[3], const float std[3], void* dst, int32_t dst_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Dense contractions on MFMA  ([d2] Conv2d / FrozenBatchNorm2d (folded) / nn.Linear of build_resnet_fpn_backbone
 * (Base-RCNN-FPN.yaml:3-8), ClsFreeRPNHead.conv (classification_free_rpn.py:158), FastRCNNConvFCHead
 * (osrcnn_roi_heads.py:308)).  Implicit GEMM:  out[n,oh,ow,co] = act( sum_{kh,kw,ci} in[n,oh*sh-ph+kh,
 * ow*sw-pw+kw,ci] * w[co,kh,kw,ci] + bias[co] (+ residual) ).
 * in/weight: fp16 or bf16; accumulate fp32; out: fp16/bf16/fp32. cin must be a multiple of 32, cout of 8.
 * A fully connected layer is the 1x1 case with hi=rows, wi=1.
 * res_mode: 0 none; 1 residual[n,oh,ow,co] (bottleneck shortcut); 2 residual[n,oh/2,ow/2,co]
 * (FPN top-down nearest-2x upsample-add); 3 residual[n,oh,ow,co] is a forward activation used as a ReLU mask
 * (out = residual > 0 ? value : 0; the backward-data pass, see osr_conv2d_wgrad below).
 * Strides are in ELEMENTS; the channel stride is 1.
 * pad_mode: 0 = bounds-checked zero padding; 1 = the input buffer already holds the halo (stem view).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct osr_conv_params {
    int32_t n, hi, wi, cin;
    int32_t ho, wo, cout;
    int32_t kh, kw, stride_h, stride_w, pad_h, pad_w;
    int64_t in_stride_n, in_stride_h, in_stride_w;
    int64_t out_stride_n, out_stride_h, out_stride_w;
    int64_t res_stride_n, res_stride_h, res_stride_w;
    int32_t relu;
    int32_t res_mode;
    int32_t pad_mode;
    int32_t in_dtype;  /* osr_dtype of in, weight, residual */
    int32_t out_dtype; /* osr_dtype of out */
    int32_t concurrency; /* scheduling hint: number of streams of the caller that launch onto the GPU at the same time (0 or 1:
                          * this launch has the GPU to itself). Tile selection only; results do not depend on it. */
    void* workspace;         /* optional (may be null), caller-owned, 16-byte aligned: with at least                      */
    int64_t workspace_bytes; /* osr_conv2d_fwd_workspace_bytes(p) bytes the partial last dispatch round of a deep-K 1x1 / FC
                              * layer is cut along K (fixed-order fp32 partial sums: bitwise reproducible, but not bitwise
                              * equal to the un-split sum) */
    const int32_t* row_seg_counts; /* optional (may be null), device memory: the output rows (n*ho*wo of them) come in segments */
    int32_t row_seg_rows;          /* of row_seg_rows rows of which only the first row_seg_counts[s] carry data (a padded per-image
                                    * proposal list: the box head's FC layers). A tile of output rows without any such row is
                                    * skipped and its rows are left unwritten; all other rows are computed as usual. */
} osr_conv_params;

/* Workspace with which osr_conv2d_fwd splits the tail round of this layer along K; 0 when the layer does not qualify (then a
 * null workspace costs nothing). Host-side arithmetic only. */
int64_t osr_conv2d_fwd_workspace_bytes(const osr_conv_params* p);
/* How osr_conv2d_fwd covers this layer: the tile shape of a single launch, or full rounds + split-K tail + reduction
 * (has_workspace != 0). Writes a short text ("256x256/2 rows [0,65536) + split-K x5 tail rows [65536,68368) + reduce") and
 * returns its length; host-side only (diagnostics, tests). */
int32_t osr_conv2d_fwd_describe(const osr_conv_params* p, int32_t has_workspace, char* buf, int32_t buf_bytes);

osr_status osr_conv2d_fwd(const osr_conv_params* p, const void* in, const void* weight, const float* bias,
                          const void* residual, void* out, void* stream);
/* osr_conv2d_fwd followed, in the same epilogue, by a ReLU mask: out = mask > 0 ? (conv + bias [+ residual]) : 0.
 * `mask` has in_dtype and out's logical layout (addressed with out_stride_*). This is the join of a residual block in the
 * backward pass -- data gradient of one branch + the gradient of the other (res_mode 1), through the ReLU of the layer
 * below ([d2] BottleneckBlock.forward's `F.relu_(out + shortcut)`, entered from train.py:145 `losses.backward()`) -- in one
 * launch. res_mode 0 or 1. Returns OSR_ERR_UNSUPPORTED (nothing launched) outside the BK=64 kernel's envelope
 * (cin % 64 != 0): run osr_conv2d_fwd + osr_relu_mask instead. */
osr_status osr_conv2d_fwd_masked(const osr_conv_params* p, const void* in, const void* weight, const float* bias,
                                 const void* residual, const void* mask, void* out, void* stream);

/* A bottleneck's 3x3 convolution and the 1x1 convolution behind it in ONE launch ([d2] BottleneckBlock.forward: conv2 -> conv3 ->
 * `out += shortcut; relu`, /root/reference/configs/Base-RCNN-FPN.yaml:3-8):
 *     out = relu(conv1x1(act(conv(in, weight) + bias), w3) + bias3 + residual),   act = ReLU when p->relu.
 * p describes the FIRST convolution (its output, cout channels, never reaches HBM; out_dtype == in_dtype f16/bf16; res_mode,
 * out_stride_* and row_seg_* are ignored); w3 is [cout3][cout] in the same dtype, bias3 fp32; residual and out are dense
 * (n*ho*wo, cout3) tensors of that dtype. Same K order and rounding points as osr_conv2d_fwd twice: bit-identical results.
 * Fused shapes: cout == 128, cout3 == 512 (the res3 blocks); anything else returns OSR_ERR_UNSUPPORTED, nothing launched. */
osr_status osr_conv2d_chain_fwd(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* w3,
                                const float* bias3, int32_t cout3, const void* residual, void* out, void* stream);
/* The same, also writing the first convolution's activated output (what the separate launch would have stored), dense
 * (n*ho*wo, cout) rows in the storage dtype, to mid_out (nullable): the training step keeps it for the block's backward (ReLU mask of
 * conv3's data gradient, input of conv3's weight gradient) and still saves the second launch and its re-read. */
osr_status osr_conv2d_chain_fwd_ex(const osr_conv_params* p, const void* in, const void* weight, const float* bias, const void* w3,
                                   const float* bias3, int32_t cout3, const void* residual, void* out, void* mid_out,
                                   void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * One whole ResNet bottleneck block in ONE launch: y = relu(conv3(relu(conv2(relu(conv1(x))))) + shortcut(x)),
 * 1x1 -> 3x3 (pad 1) -> 1x1, stride 1, FrozenBN folded into weights / biases ([d2] BottleneckBlock.forward, built by
 * build_resnet_fpn_backbone for /root/reference/configs/Base-RCNN-FPN.yaml:3-8). The two cmid-channel intermediates never
 * leave the chip: the block reads x once and writes y once (the three separate convolutions of a res2 block are HBM-bound
 * and move twice the bytes). Fused shapes: the res2 blocks -- cmid 64, cout 256, and either cin 256 with the identity
 * shortcut (has_proj 0) or cin 64 with a 1x1 projection shortcut wsc / bsc (has_proj 1). Anything else returns
 * OSR_ERR_UNSUPPORTED with nothing launched: run osr_conv2d_fwd three (four) times instead.
 * in (n,h,w,cin), out (n,h,w,cout) NHWC contiguous, dtype f16/bf16; weights packed [cout][kh][kw][cin] in the same dtype,
 * biases fp32. Same K order and the same rounding points as the separate launches; with has_proj the shortcut's output
 * is accumulated in fp32 with conv3 instead of being rounded to the storage dtype first (one rounding fewer).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct osr_bottleneck_params {
    int32_t n, h, w;
    int32_t cin, cmid, cout;
    int32_t dtype;    /* osr_dtype of in, out and the weights */
    int32_t has_proj; /* 1: projection shortcut (wsc, bsc); 0: identity (cin == cout) */
} osr_bottleneck_params;
osr_status osr_bottleneck_fwd(const osr_bottleneck_params* p, const void* in, const void* w1, const float* b1,
                              const void* w2, const float* b2, const void* w3, const float* b3, const void* wsc,
                              const float* bsc, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * ResizeShortestEdge on the device for uint8 images: [d2] ResizeTransform.apply_image = PIL Image.resize(BILINEAR)
 * (INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST, /root/reference/configs/Base-RCNN-FPN.yaml:43; the loader the reference builds
 * at train.py:129 does it on the host). Pillow's algorithm (Resample.c): separable triangle filter whose support grows
 * with the down-scaling factor, horizontal pass into an 8-bit intermediate, then the vertical pass, 22-bit fixed-point
 * coefficients. The caller computes the tables on the host from the two sizes (Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc: bounds (n,2) = first input index and tap count per output index, coef (n,k) int32); the result
 * is the PIL image bit for bit.
 * in: (h, w, 3) uint8 interleaved (row stride in bytes); out: (3, nh, nw) uint8 planar -- the "image" tensor of a model
 * input dict; tmp: osr_resize_tmp_bytes(y_rows, nw) bytes, rows [y_first, y_first + y_rows) = the input rows the vertical
 * pass reads (Pillow resamples only those horizontally).
 * --------------------------------------------------------------------------------------------------------- */
int64_t osr_resize_tmp_bytes(int32_t rows, int32_t nw);
osr_status osr_resize_bilinear_u8(const uint8_t* in, int32_t h, int32_t w, int64_t in_row_stride, const int32_t* xbounds,
                                  const int32_t* xcoef, int32_t kx, const int32_t* ybounds, const int32_t* ycoef,
                                  int32_t ky, int32_t y_first, int32_t y_rows, int32_t nh, int32_t nw, uint8_t* tmp,
                                  int64_t tmp_bytes, uint8_t* out, void* stream);

/* The whole ResNet stem in ONE launch (csrc/osr_stem_pool.hip): [d2] BasicStem.forward = conv1 (7x7, stride 2, pad 3, 3 -> 64,
 * FrozenBN folded) -> ReLU -> F.max_pool2d(3, 2, 1), built by build_resnet_fpn_backbone (/root/reference/configs/Base-RCNN-FPN.yaml:3-8).
 * xpad: osr_preprocess' (n, hp + 6, osr_stem_padded_width(wp), 4) image; w_view: the stem view (64, w_rows, 1, 32) of
 * osr_conv2d_fwd's stem path (w_rows 7 or 8; row ky holds 8 taps x 4 channels, the 8th tap and the 4th channel zero); out:
 * (n, hp / 4, wp / 4, 64) (odd halves round up as the two layers do). The 64-channel stem output never reaches HBM. Same K order and
 * rounding points as osr_conv2d_fwd(stem view) followed by osr_maxpool3x3s2. dtype f16 / bf16 (hp, wp even). */
osr_status osr_stem_maxpool_fwd(const void* xpad, int32_t n, int32_t hp, int32_t wp, const void* w_view, int32_t w_rows, const float* bias,
                                void* out, int32_t dtype, void* stream);
/* The same from the RAW batch: osr_preprocess' normalisation, halo and /32 padding are applied while the kernel stages its input patch
 * (src (n,3,h,w) NCHW uint8 or float32, mean / std as osr_preprocess): same bits, no pre-padded copy of the batch in HBM. */
osr_status osr_stem_maxpool_fwd_raw(const void* src, int32_t src_is_u8, int32_t n, int32_t h, int32_t w, int32_t hp, int32_t wp,
                                    const float mean[3], const float std[3], const void* w_view, int32_t w_rows, const float* bias,
                                    void* out, int32_t dtype, void* stream);
/* [d2] F.max_pool2d(k=3,s=2,p=1) of the ResNet stem, NHWC contiguous. */
osr_status osr_maxpool3x3s2(const void* in, int32_t n, int32_t hi, int32_t wi, int32_t c, void* out, int32_t dtype,
                            void* stream);
/* [d2] LastLevelMaxPool: p6 = max_pool2d(p5, k=1, s=2) = stride-2 subsample, NHWC contiguous. */
osr_status osr_subsample2(const void* in, int32_t n, int32_t hi, int32_t wi, int32_t c, void* out, int32_t dtype,
                          void* stream);

/* fp32 GEMM out[m,n] = a[m,k] * w[n,k]^T + bias[n] on the exact-f32 MFMA (PLN encoder/decoder,
 * prototype_learning_network.py:204-205; cls_score, softmax_classifier.py:306). lda/ldo in elements. */
osr_status osr_gemm_f32(const float* a, int64_t lda, const float* w, const float* bias, float* out, int64_t ldo,
                        int32_t m, int32_t n, int32_t k, int32_t relu, void* stream);
/* fp32 out[m,n] = sum_k a[k,m] * b[k,n] on the exact-f32 MFMA: dW = dy^T x of the same layers in the training step (the
 * autograd backward of F.linear at prototype_learning_network.py:204-205, softmax_classifier.py:306), operands as they lie
 * (row = sample). The sample axis is split over workgroups when `workspace` holds osr_gemm_f32_tn_workspace_bytes(m,n,k);
 * the partial sums are added in split order. */
int64_t osr_gemm_f32_tn_workspace_bytes(int32_t m, int32_t n, int32_t k);
osr_status osr_gemm_f32_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int32_t m,
                           int32_t n, int32_t k, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * CF-RPN head tail: ClsFreeRPNHead.forward after the 3x3 conv+ReLU (classification_free_rpn.py:159-161):
 * t/max(||t||_2,1e-12) over channels, 1x1 -> 4 ltrb deltas, 1x1 -> centerness, sigmoid.
 * t: (rows, c) channels-last hidden state; w_delta (4,c), w_ctr (1,c) fp32. Outputs fp32.
 * --------------------------------------------------------------------------------------------------------- */
osr_status osr_cfrpn_head_tail(const void* t, int32_t t_dtype, int64_t rows, int32_t c, const float* w_delta,
                               const float* b_delta, const float* w_ctr, const float* b_ctr, float* deltas,
                               float* ctr, void* stream);

/* The whole ClsFreeRPNHead.forward for one pyramid level in ONE launch (classification_free_rpn.py:157-161):
 * 3x3 conv + bias + ReLU on MFMA, then -- without the hidden state ever leaving the chip -- the channel
 * L2-normalise, both 1x1 convs and the sigmoid. p describes the 3x3 convolution (cout must be 256, cin %% 64 == 0,
 * no residual; p->relu/out_* are ignored). w_tail: (5, 256) fp32, rows 0-3 = anchor_deltas, row 4 = centerness;
 * b_tail: (5). deltas/ctr are indexed by pixel (n*ho + oh)*wo + ow. Returns OSR_ERR_UNSUPPORTED outside that
 * envelope: run osr_conv2d_fwd + osr_cfrpn_head_tail instead. The tail's dot products and ||t||^2 also run on the matrix
 * cores (each tail row scaled by a power of two, then its fp32 weights as three storage-dtype terms each -- exact for every weight
 * within 2^-11 of its row's largest, at most 2^-35 of that largest off below --, fp32 accumulation, the scale undone on the
 * sums; subnormal fp16 terms are multiplied as they are): equal to the un-fused pair up to the fp32 summation order (1e-7
 * absolute on O(1) outputs), not bit for bit. */
osr_status osr_cfrpn_head_fwd(const osr_conv_params* p, const void* in, const void* weight, const float* bias,
                              const float* w_tail, const float* b_tail, float* deltas, float* ctr, void* stream);
/* TWO convolutions of the same input and geometry in ONE launch: [d2] BottleneckBlock.forward of a stage's first block applies
 * `self.conv1` (1x1, stride s, FrozenBN folded, ReLU) and `self.shortcut` (1x1, stride s, no ReLU) to the same x
 * (build_resnet_fpn_backbone selected by /root/reference/configs/Base-RCNN-FPN.yaml:3-8). p gives the shared geometry (n, hi, wi, cin,
 * ho, wo, kernel, stride, pad, input strides, dtypes); its cout / relu / out strides / residual fields are ignored. Convolution A
 * (w_a (cout_a, kh, kw, cin), bias_a, relu_a) writes the dense (n, ho, wo, cout_a) tensor out_a, B likewise; cout_a and cout_b are
 * multiples of 128, cin of 64, storage f16 / bf16. Bit-identical to two osr_conv2d_fwd launches that run on the 128 x 128 tile.
 * Returns OSR_ERR_UNSUPPORTED (nothing launched) outside that envelope. */
osr_status osr_conv2d_fwd_pair(const osr_conv_params* p, const void* in, const void* w_a, const float* bias_a, int32_t cout_a,
                               int32_t relu_a, void* out_a, const void* w_b, const float* bias_b, int32_t cout_b, int32_t relu_b,
                               void* out_b, void* stream);

/* One pyramid level of a multi-level launch: a dense NHWC input (n, hi, wi, cin) of the storage dtype and where its results go. */
typedef struct osr_conv_level {
    const void* in;   /* (n, hi, wi, cin), dense */
    void* out;        /* osr_conv2d_fwd_levels: (n, hi, wi, cout), dense, storage dtype. osr_cfrpn_head_fwd_levels: the hidden state
                       * t = relu(conv + bias), (n*hi*wi, 256) rows, or NULL */
    float* deltas;    /* osr_cfrpn_head_fwd_levels: (n*hi*wi, 4) ltrb deltas (osr_conv2d_fwd_levels: ignored) */
    float* ctr;       /* osr_cfrpn_head_fwd_levels: (n*hi*wi) centerness */
    const void* weight; /* the level's own (cout, kh, kw, cin) weights and (cout) fp32 bias -- the FPN has one output conv per level --, */
    const float* bias;  /* or NULL: the shared `weight` / `bias` arguments of the call */
    int32_t n, hi, wi;
    int32_t reserved;
} osr_conv_level;
#define OSR_MAX_CONV_LEVELS 6
/* Stride-1, same-padding K x K convolutions of ONE shape (kernel size, cin, cout) applied to several feature maps in ONE launch, each
 * level with its own weights (osr_conv_level.weight / .bias) or the shared `weight` / `bias` (may be NULL when every level brings its
 * own): [d2] FPN.forward's `output_conv(prev_features)` per level (the backbone build_resnet_fpn_backbone selected by
 * /root/reference/configs/Base-RCNN-FPN.yaml:3-8). p gives kh == kw (odd), pad == kh / 2, stride 1, cin, cout (multiple of 256),
 * dtypes (f16 / bf16 in and out) and relu; its n / hi / wi / ho / wo / strides / residual fields are ignored: every level is a dense
 * tensor described by its osr_conv_level. Outputs are bit-identical to osr_conv2d_fwd per level whenever that picks its 256 x 256 tile
 * (and equal within fp32 summation order otherwise: the split-K tail of a single-level launch adds its K ranges separately).
 * Why: at batch 16 the levels p2..p5 are 4200 + 1050 + 263 + 66 tiles of 256 x 256 rows x channels; launched one by one each level pays
 * its own partial last dispatch round on 256 CUs and the small levels leave most of the chip idle; one grid pays one. */
osr_status osr_conv2d_fwd_levels(const osr_conv_params* p, int32_t nlevels, const osr_conv_level* levels, const void* weight,
                                 const float* bias, void* stream);
/* ClsFreeRPNHead.forward over ALL pyramid levels in ONE launch (classification_free_rpn.py:157-161: `for x in features:` applies the
 * same conv / anchor_deltas / centerness weights to every level): osr_cfrpn_head_fwd's fused kernel on the 256-row tile with a level
 * table. Same envelope as osr_cfrpn_head_fwd (cout == 256, cin %% 64 == 0, >= 8 K slices); bit-identical to it per level whenever it
 * runs its 256-row tile (levels of >= 512 tiles), equal within fp32 summation order to its 128-row tile otherwise. */
osr_status osr_cfrpn_head_fwd_levels(const osr_conv_params* p, int32_t nlevels, const osr_conv_level* levels, const void* weight,
                                     const float* bias, const float* w_tail, const float* b_tail, void* stream);
/* The same, also writing the hidden state t = relu(conv + bias) in the storage dtype, (n*ho*wo, 256) rows (hidden_out may be
 * NULL): the training step keeps it for the head's backward (osr_cfrpn_tail_bwd, the 3x3 conv's weight gradient) instead of
 * running the un-fused pair that writes t and reads it back. */
osr_status osr_cfrpn_head_fwd_ex(const osr_conv_params* p, const void* in, const void* weight, const float* bias,
                                 const float* w_tail, const float* b_tail, float* deltas, float* ctr, void* hidden_out,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Proposal selection: ClsFreeRPN.predict_proposals -> _decode_proposals (classification_free_rpn.py:558-610)
 * + find_top_rpn_proposals (find_top_proposals.py:60-127) for all images and levels in two launches.
 * Level l holds ctr[l_off + img*hw_l*a + i] and deltas[(same)*4]. Per (image, level): stable top-k of
 * centerness (value-descending, lower index first), ltrb decode of the k survivors around their anchors
 * ([d2] DefaultAnchorGenerator + Box2BoxTransformLinear), finite filter, clip to the image, drop empty boxes;
 * levels concatenated level-major (no NMS, no cross-level re-sort: the reference has both commented out).
 * Outputs are padded to cap = osr_rpn_select_capacity(): boxes (n,cap,4), scores (n,cap), src_index (n,cap)
 * (index into the image's concatenated anchor list), batch_idx (n*cap: image id, -1 for padding), counts (n).
 * status_flags[0] is set non-zero when any non-finite prediction was met (training raises on it).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct osr_rpn_levels {
    int32_t num_levels;
    int32_t num_anchors;            /* A: cell anchors per location */
    int32_t h[OSR_MAX_LEVELS];
    int32_t w[OSR_MAX_LEVELS];
    int32_t stride[OSR_MAX_LEVELS];
    int64_t offset[OSR_MAX_LEVELS]; /* element offset of level l in ctr (x4 in deltas) */
} osr_rpn_levels;

int32_t osr_rpn_select_capacity(const osr_rpn_levels* lv, int32_t pre_nms_topk);
int64_t osr_rpn_select_workspace_bytes(const osr_rpn_levels* lv, int32_t n, int32_t pre_nms_topk);
osr_status osr_rpn_select(const osr_rpn_levels* lv, const float* cell_anchors /* (L,A,4) */, const float* ctr,
                          const float* deltas, int32_t n, const int32_t* image_hw /* (n,2) */, int32_t pre_nms_topk,
                          float min_box_size, float* boxes, float* scores, int32_t* src_index, int32_t* batch_idx,
                          int32_t* counts, int32_t* status_flags, void* workspace, int64_t workspace_bytes,
                          void* stream);

/* osr_rpn_select with the decode rule and an extra output selectable -- what the stock detectron2 RPN of
 * /root/reference/configs/Base-RCNN-FPN.yaml:9-21 (BASELINE config 1) needs next to the CF-RPN:
 * decode_mode 0 = [d2] Box2BoxTransformLinear (ltrb, the CF-RPN; reg_weights ignored), 1 = [d2] Box2BoxTransform with
 * reg_weights (dx,dy,dw,dh; dw/dh clamped at log(1000/16)) -- MODEL.RPN.BBOX_REG_WEIGHTS. `ctr` is then the objectness LOGIT
 * (top-k on the raw value, as [d2] find_top_rpn_proposals does). level_out (nullable): (n,cap) pyramid level of every kept
 * slot (-1 padding): the category of the per-level batched NMS that follows (osr_nms_topk, thr MODEL.RPN.NMS_THRESH). */
osr_status osr_rpn_select_ex(const osr_rpn_levels* lv, const float* cell_anchors, const float* ctr, const float* deltas,
                             int32_t n, const int32_t* image_hw, int32_t pre_nms_topk, float min_box_size,
                             int32_t decode_mode, const float reg_weights[4], float* boxes, float* scores,
                             int32_t* src_index, int32_t* batch_idx, int32_t* level_out, int32_t* counts,
                             int32_t* status_flags, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * RoIAlign over the pyramid: [d2] ROIPooler.forward (level = floor(4+log2(sqrt(area)/224+1e-8)) clamped to
 * [min,max]) -> torchvision roi_align(aligned=True, sampling_ratio=0) (osrcnn_roi_heads.py:108-113,306).
 * feats[l]: (n, h_l, w_l, c) NHWC. boxes (m,4) fp32, batch_idx (m) (<0 => row of zeros).
 * out: (m, p, p, c) i.e. out[m][(ph*p+pw)*c + ch]  (the reference's (m,c,p,p) is the same data permuted;
 * the FC1 weight is permuted once at load time instead).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct osr_pyramid {
    int32_t num_levels;
    int32_t c;
    int32_t h[OSR_MAX_LEVELS];
    int32_t w[OSR_MAX_LEVELS];
    float scale[OSR_MAX_LEVELS];
    const void* data[OSR_MAX_LEVELS];
} osr_pyramid;

osr_status osr_roi_align_fwd(const osr_pyramid* feats, int32_t feat_dtype, int32_t n, const float* boxes,
                             const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                             int32_t canonical_size, int32_t min_level, void* out, int32_t out_dtype, void* stream);

/* The same with an explicit processing order: `order` is a permutation of 0..m-1 (or NULL = list order); workgroup i pools RoI
 * order[i] into ITS OWN row out[order[i]], so the result is bit-identical for every order. `order_nvalid` (device pointer, may
 * be NULL): the number of leading entries of `order` that are real RoIs, the rest being padding rows (batch index -1): the
 * kernel then gives every XCD the same share of the real work. osr_roi_locality_order fills both so that RoIs that are
 * neighbours in the image are neighbours in time on one XCD (bucket sort by image, pyramid level and 32x32-pixel tile of the box
 * centre; the level rule is the one of osr_roi_align_fwd; padding rows last): the proposals of an image overlap each other
 * several times over, and in score order every overlap is a re-read from HBM.
 * workspace: osr_roi_locality_order_workspace_bytes(n, m) bytes. */
osr_status osr_roi_align_fwd_ordered(const osr_pyramid* feats, int32_t feat_dtype, int32_t n, const float* boxes,
                                     const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                     int32_t canonical_size, int32_t min_level, const int32_t* order,
                                     const int32_t* order_nvalid, void* out, int32_t out_dtype, void* stream);
/* osr_roi_align_fwd_ordered with flags. OSR_ROI_NO_PADDING_FILL: padding rows (batch index -1) are left UNWRITTEN instead of
 * zero-filled -- for a caller whose consumers never read them (the engine: the box head skips or ignores those rows; a fifth of the
 * benchmark's list, 0.37 GB of zeros per step). */
#define OSR_ROI_NO_PADDING_FILL 1
osr_status osr_roi_align_fwd_ordered_ex(const osr_pyramid* feats, int32_t feat_dtype, int32_t n, const float* boxes,
                                        const int32_t* batch_idx, int64_t m, int32_t pooled, int32_t canonical_level,
                                        int32_t canonical_size, int32_t min_level, const int32_t* order,
                                        const int32_t* order_nvalid, int32_t flags, void* out, int32_t out_dtype, void* stream);
int64_t osr_roi_locality_order_workspace_bytes(int32_t n, int64_t m);
osr_status osr_roi_locality_order(const osr_pyramid* feats, int32_t n, const float* boxes, const int32_t* batch_idx, int64_t m,
                                  int32_t canonical_level, int32_t canonical_size, int32_t min_level, int32_t* order,
                                  int32_t* nvalid, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Box predictor tail: OpensetFastRCNNOutputLayers.forward + predict_boxes + predict_ious
 * (osrcnn_fast_rcnn.py:262-263,423,443-446) and the per-row part of fast_rcnn_inference_single_image
 * (:109-126): deltas = x W_b^T + b, iou = sigmoid(x w_i + b), box = Box2BoxTransform.apply_deltas (weights
 * wx,wy,ww,wh; scale clamp log(1000/16)), score = sqrt(iou*ctr) (mean_type 0) or (iou+ctr)/2 (1), finite filter,
 * clip, score > thresh. x: (m, k) fp32 box-head features. w: (5,k): rows 0-3 bbox_pred, row 4 iou_pred.
 * Outputs: pred_deltas (m,4) raw, pred_iou (m), boxes (m,4) clipped, score (m), cand (m) int32 0/1.
 * --------------------------------------------------------------------------------------------------------- */
osr_status osr_box_predictor_tail(const float* x, int64_t m, int32_t k, const float* w, const float* b,
                                  const float* proposals, const float* ctr, const int32_t* batch_idx,
                                  const int32_t* image_hw, const float reg_weights[4], int32_t mean_type,
                                  float score_thresh, float* pred_deltas, float* pred_iou, float* boxes,
                                  float* score, int32_t* cand, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Segmented stable sort + greedy per-class NMS + top-k: [d2] detectron2.layers.batched_nms -> torchvision
 * nms (osrcnn_fast_rcnn.py:135-137; softmax_classifier.py:93-95,154-156), vanilla per-class semantics:
 * candidates ordered by score descending (ties: lower index first); a candidate is suppressed by an already
 * kept one of the same class when inter/(a_i+a_j-inter) > thr; the first `topk` kept are returned, in order.
 * thr >= 1 suppresses nothing (pure sort + top-k). Segment s covers elements
 * [s*seg_stride, s*seg_stride + seg_len[s]) of boxes/scores/cls/cand; cand==0 elements are skipped.
 * keep: (num_segments, topk) indices relative to the segment start; keep_count: (num_segments).
 * --------------------------------------------------------------------------------------------------------- */
int64_t osr_nms_topk_workspace_bytes(int32_t num_segments, int64_t seg_stride);
osr_status osr_nms_topk(const float* boxes, const float* scores, const int32_t* cls /* nullable: one class */,
                        const int32_t* cand /* nullable: all */, int32_t num_segments, int64_t seg_stride,
                        const int32_t* seg_len, float thr, int32_t topk, int32_t* keep, int32_t* keep_count,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Gather rows: dst[s, j, :] = src[s*seg_stride + keep[s, j], :] for j < keep_count[s], else 0
 * (boxes[keep], scores[keep], feats[keep] at osrcnn_fast_rcnn.py:138). row_elems fp32 elements per row. */
osr_status osr_gather_rows(const float* src, int64_t seg_stride, int32_t row_elems, const int32_t* keep,
                           const int32_t* keep_count, int32_t num_segments, int32_t topk, float* dst, void* stream);

/* Row-wise L2 normalisation x/max(||x||,1e-12) (F.normalize; prototype_learning_network.py:199). */
osr_status osr_l2_normalize_rows(const float* x, int32_t rows, int32_t d, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * PLN inference tail (prototype_learning_network.py:206-223, COS distance): normalise each embedding, cosine
 * distance to the (already normalised) prototypes, min over reps then min/argmin over classes (ties: lower
 * class), unknown if min > unk_thr. class_map (nullable, (num_known)): GraspNet class_id remap (:222).
 * emb: (rows, d). rows_valid (nullable): per-segment valid counts for padded (segments, seg_rows) layouts;
 * padded rows get class -1. Outputs: pred_class (rows) int64, min_dist (rows).
 * --------------------------------------------------------------------------------------------------------- */
osr_status osr_pln_tail(const float* emb, int64_t rows, int32_t d, const float* protos_normed, int32_t num_known,
                        int32_t reps, float unk_thr, int64_t unknown_id, const int64_t* class_map,
                        const int32_t* rows_valid, int32_t seg_rows, int64_t* pred_class, float* min_dist,
                        void* stream);
/* MODEL.PLN.DISTANCE_TYPE (prototype_learning_network.py:155-160, 213-218): the distance between the normalised embedding and
 * the normalised prototypes. osr_pln_tail / osr_pln_loss_fwd / osr_pln_loss_bwd are the COS forms with one prototype per class. */
enum { OSR_PLN_DIST_COS = 0, OSR_PLN_DIST_L1 = 1, OSR_PLN_DIST_L2 = 2 };
osr_status osr_pln_tail_ex(const float* emb, int64_t rows, int32_t d, const float* protos_normed, int32_t num_known,
                           int32_t reps, int32_t distance_type, float unk_thr, int64_t unknown_id,
                           const int64_t* class_map, const int32_t* rows_valid, int32_t seg_rows, int64_t* pred_class,
                           float* min_dist, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Softmax classifier candidates: SoftMaxClassifier.inference up to the NMS calls (softmax_classifier.py:300-307,
 * 66-88, 126-149). Per image (segment of seg_rows detections, det_count[s] valid): known detections
 * (pred_class != unknown_id): softmax over num_known+1 logits, drop the background column, candidate for every
 * (det, class) with prob > known_thresh in row-major order; unknown detections: candidate if objectness score >
 * unknown_thresh. Candidate arrays are padded per image: known cap = seg_rows*num_known, unknown cap = seg_rows.
 * Outputs (k = known, u = unknown): *_boxes (n,cap,4), *_scores (n,cap), k_cls (n,cap) int32, *_det (n,cap)
 * int32 source detection, *_count (n).
 * --------------------------------------------------------------------------------------------------------- */
osr_status osr_softmax_candidates(const float* logits, int32_t num_known, const float* det_boxes,
                                  const float* det_scores, const int64_t* pred_class, const int32_t* det_count,
                                  int32_t n, int32_t seg_rows, int64_t unknown_id, float known_thresh,
                                  float unknown_thresh, float* k_boxes, float* k_scores, int32_t* k_cls, int32_t* k_det,
                                  int32_t* k_count, float* u_boxes, float* u_scores, int32_t* u_det, int32_t* u_count,
                                  void* stream);

/* [d2] FastRCNNOutputLayers.inference -> fast_rcnn_inference_single_image up to (not including) the NMS, for the stock
 * StandardROIHeads of /root/reference/configs/Base-RCNN-FPN.yaml:22-28 (BASELINE config 1). logits (n*seg_rows, K+1),
 * deltas (n*seg_rows, R*4) with R = num_bbox_reg_classes (K, or 1 when class-agnostic), prop_boxes (n, seg_rows, 4) padded with
 * prop_count (n) valid rows. Per valid row: softmax; Box2BoxTransform(reg_weights) decode of every class's box; the row is
 * dropped when any box coordinate or probability is non-finite; boxes clipped to image_hw; every (row, class < K) with
 * p > score_thresh becomes a candidate, row-major. Outputs padded to seg_rows*K per image: c_boxes (n,cap,4), c_scores,
 * c_cls, c_row (proposal row of the candidate), c_count (n). Follow with osr_nms_topk(cls = c_cls, thr NMS_THRESH_TEST,
 * topk DETECTIONS_PER_IMAGE). */
osr_status osr_fastrcnn_candidates(const float* logits, const float* deltas, int32_t num_classes,
                                   int32_t num_bbox_reg_classes, const float* prop_boxes, const int32_t* prop_count,
                                   int32_t n, int32_t seg_rows, const int32_t* image_hw, const float reg_weights[4],
                                   float score_thresh, float* c_boxes, float* c_scores, int32_t* c_cls, int32_t* c_row,
                                   int32_t* c_count, void* stream);

/* Final assembly: output order [unknown..., known...] (softmax_classifier.py:328-334), class ids int64
 * (unknown_id for the unknown group, class_map[c] or c for known). out_* padded to (n, u_topk + k_topk). */
osr_status osr_assemble_detections(const float* k_boxes, const float* k_scores, const int32_t* k_cls,
                                   const int32_t* k_keep, const int32_t* k_keep_count, int64_t k_stride, int32_t k_topk,
                                   const float* u_boxes, const float* u_scores, const int32_t* u_keep,
                                   const int32_t* u_keep_count, int64_t u_stride, int32_t u_topk, int32_t n,
                                   int64_t unknown_id, const int64_t* class_map, float* out_boxes, float* out_scores,
                                   int64_t* out_classes, int32_t* out_count, void* stream);

/* [d2] detector_postprocess on the device (the step after the path): per image, boxes * (scale_x, scale_y),
 * clip to (out_h, out_w), drop boxes without positive width and height, keep the order. In/out: padded (n, cap, .) + counts;
 * scale_xy (n,2) = {out_w / w, out_h / h}; out_hw (n,2) = {out_h, out_w}. Not in place. */
osr_status osr_detector_postprocess(const float* boxes, const float* scores, const int64_t* classes, const int32_t* count,
                                    int32_t n, int32_t cap, const float* scale_xy, const int32_t* out_hw, float* out_boxes,
                                    float* out_scores, int64_t* out_classes, int32_t* out_count, void* stream);

/* =========================================================================================================
 * Training step, forward half: targets and losses. Gradients are not
 * produced by this library yet; every function below is a forward kernel whose outputs equal the reference's
 * forward values. Random subsampling takes caller-supplied uniform keys: the k smallest keys of a class are
 * kept (ties: lower index) -- the role torch.randperm plays in [d2] subsample_labels -- so that runs are
 * reproducible and the selected lists are comparable bit for bit.
 * ========================================================================================================= */

/* ---------------------------------------------------------------------------------------------------------
 * Anchor <-> GT matching: ClsFreeRPN.label_and_sample_anchors up to the subsampling
 * (classification_free_rpn.py:359-371): pairwise IoU (GT x anchors), argmax over GT (first maximum),
 * Matcher(reg thresholds, labels [0,-1,1], allow_low_quality_matches) and the same for the objectness
 * thresholds. Anchors are generated in-kernel from the level table (image-major list of R = sum h*w*a).
 * gt_boxes: (n, gmax, 4) padded, gt_count (n). Images without GT get label 0 everywhere and matched_idx 0.
 * Outputs: matched_idx (n,R) int32, matched_iou (n,R), labels_reg / labels_obj (n,R) int8 in {-1,0,1}.
 * workspace: n*gmax*4 bytes.
 * --------------------------------------------------------------------------------------------------------- */
osr_status osr_rpn_match_anchors(const osr_rpn_levels* levels, const float* cell_anchors, int32_t n,
                                 const float* gt_boxes, const int32_t* gt_count, int32_t gmax, float reg_lo,
                                 float reg_hi, float obj_lo, float obj_hi, int32_t* matched_idx, float* matched_iou,
                                 int8_t* labels_reg, int8_t* labels_obj, void* workspace, int64_t workspace_bytes,
                                 void* stream);

/* [d2] subsample_labels + ClsFreeRPN._subsample_labels (classification_free_rpn.py:299-316), in place: keep
 * min(int(num_samples*positive_fraction), #pos) positives and min(num_samples - kept_pos, #neg) negatives with
 * the smallest keys, everything else becomes -1. labels, keys: (n, r). num_samples <= 512. */
osr_status osr_subsample_labels(int8_t* labels, const float* keys, int32_t n, int64_t r, int32_t num_samples,
                                float positive_fraction, int32_t* num_pos_out, int32_t* num_neg_out, void* stream);

/* Matched GT box per anchor and the centerness target (classification_free_rpn.py:386-402): ltrb deltas of the
 * anchor against its matched GT, zero unless the anchor centre lies inside, sqrt(min(l,r)/max(l,r) *
 * min(t,b)/max(t,b)), zero where the (subsampled) objectness label is 0. matched_boxes (n,R,4), ctr_target (n,R). */
osr_status osr_rpn_anchor_targets(const osr_rpn_levels* levels, const float* cell_anchors, int32_t n,
                                  const float* gt_boxes, const int32_t* gt_count, int32_t gmax,
                                  const int32_t* matched_idx, const int8_t* labels_obj, float* matched_boxes,
                                  float* ctr_target, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * ClsFreeRPN.losses, BBOX_REG_LOSS_TYPE "iou" (classification_free_rpn.py:446-490, box_regression_w_iou.py:49-61).
 * pred_deltas / pred_ctr: the level-major buffers osr_cfrpn_head_tail writes (levels->offset). Targets are
 * image-major (n, R). out6 = {loss_rpn_loc, loss_rpn_ctr, num_pos, num_neg, obj_num_pos, obj_num_neg}; both
 * losses are divided by batch_size_per_image * n and multiplied by their weight. workspace: 6 KiB.
 * --------------------------------------------------------------------------------------------------------- */
osr_status osr_rpn_losses_fwd(const osr_rpn_levels* levels, const float* cell_anchors, int32_t n,
                              const float* pred_deltas, const float* pred_ctr, const int8_t* labels_reg,
                              const int8_t* labels_obj, const float* matched_boxes, const float* ctr_target,
                              float loc_weight, float ctr_weight, int32_t batch_size_per_image, float* out6,
                              void* workspace, int64_t workspace_bytes, void* stream);
/* The loss functions the reference selects by name (box_regression_w_iou.py:13-85: BBOX_REG_LOSS_TYPE "smooth_l1" | "iou" |
 * "giou" | "diou" | "ciou" with SMOOTH_L1_BETA; classification_free_rpn.py:475-481 and osrcnn_fast_rcnn.py:368: smooth L1 with
 * its own beta for the centerness / IoU regression). NULL options = what both Openset yaml files select: "iou" for the CF-RPN
 * boxes, "smooth_l1" for the RoI boxes, every beta 0 (= L1). */
enum { OSR_BOX_LOSS_IOU = 0, OSR_BOX_LOSS_SMOOTH_L1 = 1, OSR_BOX_LOSS_GIOU = 2, OSR_BOX_LOSS_DIOU = 3, OSR_BOX_LOSS_CIOU = 4 };
typedef struct osr_loss_options {
    int32_t box_loss_type;     /* OSR_BOX_LOSS_* */
    float box_smooth_l1_beta;  /* used by OSR_BOX_LOSS_SMOOTH_L1 */
    float aux_smooth_l1_beta;  /* centerness loss (CF-RPN) / IoU regression loss (RoI head) */
} osr_loss_options;
osr_status osr_rpn_losses_fwd_ex(const osr_rpn_levels* levels, const float* cell_anchors, int32_t n,
                                 const float* pred_deltas, const float* pred_ctr, const int8_t* labels_reg,
                                 const int8_t* labels_obj, const float* matched_boxes, const float* ctr_target,
                                 float loc_weight, float ctr_weight, int32_t batch_size_per_image,
                                 const osr_loss_options* options, float* out6, void* workspace, int64_t workspace_bytes,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * OpensetROIHeads.label_and_sample_proposals (osrcnn_roi_heads.py:177-216): append the GT boxes to the
 * proposals ([d2] add_ground_truth_to_proposals, logit log((1-1e-10)/1e-10)), IoU against GT,
 * Matcher([iou_thr],[0,1]), class = GT class or num_classes (background), matched IoU, then keep batch_size
 * candidates, at most int(batch_size*positive_fraction) foreground. Candidates of image i are
 * [proposals 0..prop_count[i]), GT 0..gt_count[i])]; keys is (n, pcap+gmax): the key of proposal j is keys[i][j],
 * the key of GT g is keys[i][pcap+g], whatever the counts are.
 * Output rows per image: foreground by ascending key, then background by ascending key, padded to batch_size
 * (class -1, src -1, batch index -1; the loss kernels below skip rows with class < 0 and leave them out of their
 * normalisers, RoIAlign skips rows with batch index -1). out_src: candidate index of each row; out_batch_idx: image
 * of each row; out_counts: (n,3) = {rows, foreground, background}. batch_size <= 512.
 * --------------------------------------------------------------------------------------------------------- */
int64_t osr_roi_match_sample_workspace_bytes(int32_t n, int64_t pcap, int32_t gmax);
osr_status osr_roi_match_and_sample(const float* prop_boxes, const float* prop_logits, const int32_t* prop_count,
                                    int64_t pcap, const float* gt_boxes, const int64_t* gt_classes,
                                    const int32_t* gt_count, int32_t gmax, int32_t n, const float* keys,
                                    int32_t num_classes, int32_t batch_size, float positive_fraction, float iou_thr,
                                    float* out_boxes, float* out_logits, int64_t* out_classes, float* out_ious,
                                    float* out_gt_boxes, int32_t* out_src, int32_t* out_batch_idx, int32_t* out_counts,
                                    void* workspace, int64_t workspace_bytes, void* stream);

/* OpensetFastRCNNOutputLayers.losses (osrcnn_fast_rcnn.py:266-370): L1 between the predicted deltas and
 * Box2BoxTransform(reg_weights).get_deltas(proposal, gt), and L1 between the predicted and the matched IoU, over
 * rows with 0 <= class < num_classes; both divided by the number of rows with class >= 0 (the reference's
 * gt_classes.numel(): its row list has no padding). Predictions are read in place from the predictor GEMM output:
 * row i's deltas at pred_deltas[i*delta_stride .. +4), its IoU at pred_iou[i*iou_stride]; iou_is_logit applies the
 * sigmoid of OpensetFastRCNNOutputLayers.forward (:262) in the kernel.
 * out3 = {loss_box_reg, loss_iou, rows counted}. workspace 3 KiB. */
osr_status osr_roi_box_losses_fwd(const float* pred_deltas, int32_t delta_stride, const float* pred_iou,
                                  int32_t iou_stride, int32_t iou_is_logit, const float* proposal_boxes,
                                  const float* gt_boxes, const int64_t* gt_classes, const float* gt_iou, int64_t m,
                                  int32_t num_classes, const float reg_weights[4], float box_weight, float iou_weight,
                                  float* out3, void* workspace, int64_t workspace_bytes, void* stream);
osr_status osr_roi_box_losses_fwd_ex(const float* pred_deltas, int32_t delta_stride, const float* pred_iou,
                                     int32_t iou_stride, int32_t iou_is_logit, const float* proposal_boxes,
                                     const float* gt_boxes, const int64_t* gt_classes, const float* gt_iou, int64_t m,
                                     int32_t num_classes, const float reg_weights[4], float box_weight, float iou_weight,
                                     const osr_loss_options* options, float* out3, void* workspace,
                                     int64_t workspace_bytes, void* stream);

/* PLN.loss, COS distance, one prototype per class (prototype_learning_network.py:133-187): rows with a known
 * class and IoU > iou_thr contribute relu(d_own - alpha) + relu(beta - min d_other); the prototypes contribute
 * sum_k relu(alpha + beta - min_{j!=k} d(p_k,p_j)); total * loss_weight / (rows with class >= 0). emb: (m,d)
 * un-normalised encoder output; protos_normed (num_known, d). workspace 4 KiB. */
osr_status osr_pln_loss_fwd(const float* emb, int64_t m, int32_t d, const float* protos_normed, int32_t num_known,
                            const int64_t* gt_classes, const float* ious, float iou_thr, float alpha, float beta,
                            float loss_weight, float* out1, void* workspace, int64_t workspace_bytes, void* stream);
/* The same with REPS_PER_CLASS prototypes per class (protos_normed: (num_known * reps, d), class-major; a class's distance is
 * the minimum over its prototypes, the prototype term runs over all of them with the own-class block excluded,
 * prototype_learning_network.py:163-180) and any DISTANCE_TYPE. */
osr_status osr_pln_loss_fwd_ex(const float* emb, int64_t m, int32_t d, const float* protos_normed, int32_t num_known,
                               int32_t reps, int32_t distance_type, const int64_t* gt_classes, const float* ious,
                               float iou_thr, float alpha, float beta, float loss_weight, float* out1, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* SoftMaxClassifier.loss (softmax_classifier.py:266-285): id_map (known c -> c, num_classes -> num_known, any
 * other class is ignored), mean cross entropy over num_known+1 logits, times loss_weight. workspace 2 KiB. */
osr_status osr_softmax_ce_loss_fwd(const float* logits, int64_t m, int32_t num_known, const int64_t* gt_classes,
                                   int32_t num_classes, float loss_weight, float* out1, void* workspace,
                                   int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Backward of osr_conv2d_fwd (training step, backward half). `p` describes the FORWARD layer.
 *  - data gradient: no entry point of its own -- dx = osr_conv2d_fwd(dy, flipped/transposed weights) with stride 1,
 *    pad' = k-1-pad (host/weights.py pack_dgrad_weight); a stride-2 1x1 layer writes every second pixel of a zeroed dx
 *    through the output strides; res_mode 3 applies the ReLU mask of the layer below, res_mode 1 adds a second gradient.
 *  - weight gradient: dw[cout][kh][kw][cin] (fp32, the packed forward layout) = sum over output pixels of
 *    dy[n,oh,ow,co] * x[n, oh*sh-ph+kh, ow*sw-pw+kw, ci]; x, dy fp16/bf16 (dy dense (n,ho,wo,cout)), fp32 accumulate,
 *    split over the pixel axis with partials in the workspace and a fixed-order reduction (bitwise reproducible).
 *    accumulate != 0 adds to dw.
 *  - bias gradient: db[co] = sum_m dy[m][co]; workspace 512*cout*4 bytes.
 * --------------------------------------------------------------------------------------------------------- */
int64_t osr_conv2d_wgrad_workspace_bytes(const osr_conv_params* p);
osr_status osr_conv2d_wgrad(const osr_conv_params* p, const void* x, const void* dy, float* dw, int32_t accumulate,
                            void* workspace, int64_t workspace_bytes, void* stream);
osr_status osr_bias_grad(const void* dy, int32_t dtype, int64_t m, int32_t cout, float* db, int32_t accumulate,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training step, backward half: losses and per-row stages. Every gradient is d(sum of weighted losses)/d(tensor)
 * times `loss_scale` (static loss scaling for fp16 gradient tensors; osr_sgd_step divides it out via grad_scale).
 * --------------------------------------------------------------------------------------------------------- */

/* Gradient of osr_rpn_losses_fwd w.r.t. the head's five pre-activation outputs per anchor: d_out5 (rows, 5) level-major
 * like the predictions = {d ltrb deltas (through decode + IoU), d centerness logit (through the sigmoid)}. */
osr_status osr_rpn_losses_bwd(const osr_rpn_levels* levels, const float* cell_anchors, int32_t n,
                              const float* pred_deltas, const float* pred_ctr, const int8_t* labels_reg,
                              const int8_t* labels_obj, const float* matched_boxes, const float* ctr_target,
                              float loc_weight, float ctr_weight, int32_t batch_size_per_image, float loss_scale,
                              float* d_out5, void* stream);
osr_status osr_rpn_losses_bwd_ex(const osr_rpn_levels* levels, const float* cell_anchors, int32_t n,
                                 const float* pred_deltas, const float* pred_ctr, const int8_t* labels_reg,
                                 const int8_t* labels_obj, const float* matched_boxes, const float* ctr_target,
                                 float loc_weight, float ctr_weight, int32_t batch_size_per_image, float loss_scale,
                                 const osr_loss_options* options, float* d_out5, void* stream);

/* ClsFreeRPNHead tail backward (classification_free_rpn.py:159-161): t (rows,256) is the hidden state after the 3x3 conv's
 * ReLU; outputs dt (rows,256, same dtype, already masked by t > 0), dw_tail (5,256), db_tail (5). */
int64_t osr_cfrpn_tail_bwd_workspace_bytes(void);
osr_status osr_cfrpn_tail_bwd(const void* t, int32_t dtype, int64_t rows, const float* w_tail, const float* d_out5, void* dt,
                              float* dw_tail, float* db_tail, int32_t accumulate, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* Sparse backward of the CF-RPN head's shared 3x3 convolution. ClsFreeRPN.losses (classification_free_rpn.py:446-490) sums over
 * the sampled anchors only (:299-316: <= 2 * BATCH_SIZE_PER_IMAGE per image), so d_out5 -- and the gradient of the hidden state
 * t = relu(conv3x3(p_l)) (:159-161) -- is zero on every other anchor row; autograd runs the conv's two gradients over the dense,
 * almost-all-zero tensor. These three entry points restate them on the non-zero rows (csrc/osr_rpn_sparse.hip):
 *  osr_rpn_sparse_rows: row_ids (cap) = the rows of d_out5 (rows,5) with a non-zero entry, ascending, -1 behind them; row_map
 *    (rows) = list slot of a row or -1; count2 = {min(found, cap), found}. Rows found beyond cap are dropped (the caller sizes cap
 *    by the sampling bound and may check count2[1]).
 *  osr_rpn_gather_cols: cols (cap, 9, 256) in the feature dtype = the im2col row of every listed anchor (tap-major, the 3x3 conv's
 *    K order; zero outside the map and for the -1 slots) from its level of `feats` (levels as in lv: level-major rows
 *    (img * h + y) * w + x), and d_out5_rows (cap, 5) = its five gradients. With the head's weight W viewed as (256, 2304):
 *    t_rows = relu(cols . W^T + b), dW = dt_rows^T . cols, y = dt_rows . W.
 *  osr_rpn_scatter_cols_add: grads[l] (n, h_l, w_l, 256), in place: every pixel adds, in tap order and in fp32, the rows
 *    y[row_map[q - tap offset]][tap] of its (<= 9) listed neighbours to the value already there and rounds once (no atomics; a pixel
 *    that no listed anchor reaches is not touched). y: (cap, 9, 256) fp32. */
int64_t osr_rpn_sparse_rows_workspace_bytes(void);
osr_status osr_rpn_sparse_rows(const float* d_out5, int64_t rows, int32_t cap, int32_t* row_ids, int32_t* row_map,
                               int32_t* count2, void* workspace, int64_t workspace_bytes, void* stream);
osr_status osr_rpn_gather_cols(const osr_rpn_levels* lv, const osr_pyramid* feats, int32_t feat_dtype, int32_t n,
                               const int32_t* row_ids, int32_t cap, const float* d_out5, void* cols, float* d_out5_rows,
                               void* stream);
osr_status osr_rpn_scatter_cols_add(const osr_rpn_levels* lv, int32_t n, const int32_t* row_map, const float* y,
                                    void* const* grads, int32_t grad_dtype, void* stream);

/* Gradient of osr_roi_box_losses_fwd w.r.t. the (m,5) predictor output {4 deltas, IoU logit}. workspace 16 bytes. */
osr_status osr_roi_box_losses_bwd(const float* pred5, const float* proposal_boxes, const float* gt_boxes,
                                  const int64_t* gt_classes, const float* gt_iou, int64_t m, int32_t num_classes,
                                  const float reg_weights[4], float box_weight, float iou_weight, float loss_scale,
                                  float* d_pred5, void* workspace, int64_t workspace_bytes, void* stream);
osr_status osr_roi_box_losses_bwd_ex(const float* pred5, const float* proposal_boxes, const float* gt_boxes,
                                     const int64_t* gt_classes, const float* gt_iou, int64_t m, int32_t num_classes,
                                     const float reg_weights[4], float box_weight, float iou_weight, float loss_scale,
                                     const osr_loss_options* options, float* d_pred5, void* workspace,
                                     int64_t workspace_bytes, void* stream);

/* Gradient of osr_softmax_ce_loss_fwd w.r.t. the logits (m, num_known+1). workspace 16 bytes. */
osr_status osr_softmax_ce_loss_bwd(const float* logits, int64_t m, int32_t num_known, const int64_t* gt_classes,
                                   int32_t num_classes, float loss_weight, float loss_scale, float* d_logits,
                                   void* workspace, int64_t workspace_bytes, void* stream);

/* Gradient of osr_pln_loss_fwd w.r.t. the embeddings (m,d) and the RAW prototypes (num_known,d) (the forward normalises
 * them, prototype_learning_network.py:136). */
int64_t osr_pln_loss_bwd_workspace_bytes(int64_t m);
osr_status osr_pln_loss_bwd(const float* emb, int64_t m, int32_t d, const float* protos_raw, int32_t num_known,
                            const int64_t* gt_classes, const float* ious, float iou_thr, float alpha, float beta,
                            float loss_weight, float loss_scale, float* d_emb, float* d_protos, int32_t accumulate_protos,
                            void* workspace, int64_t workspace_bytes, void* stream);
osr_status osr_pln_loss_bwd_ex(const float* emb, int64_t m, int32_t d, const float* protos_raw, int32_t num_known,
                               int32_t reps, int32_t distance_type, const int64_t* gt_classes, const float* ious,
                               float iou_thr, float alpha, float beta, float loss_weight, float loss_scale, float* d_emb,
                               float* d_protos, int32_t accumulate_protos, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* RoIAlign backward: d feature pyramid (fp32 NHWC per level, zero-initialised by the caller; `dfeat->data` are written)
 * += scatter of dout (m,P,P,c) with the forward's geometry; fp32 atomic adds. */
osr_status osr_roi_align_bwd(const osr_pyramid* dfeat, int32_t n, const float* boxes, const int32_t* batch_idx, int64_t m,
                             int32_t pooled, int32_t canonical_level, int32_t canonical_size, int32_t min_level,
                             const void* dout, int32_t dout_dtype, void* stream);

/* The same gradient in pixel-centric form: one workgroup per 8 x 8 pixel tile of a level GATHERS the contributions of the image's RoIs
 * -- no atomics, no zero-initialised output (every element of dfeat is written exactly once, zeros where no RoI reaches), bitwise
 * reproducible (fixed summation order: list order, bin row, bin column). The RoI list must be image-major with a fixed stride:
 * rows [b * rois_per_image, (b + 1) * rois_per_image) belong to image b (batch_idx == b) or are padding (batch_idx < 0);
 * m == n * rois_per_image, rois_per_image <= 1024, c <= 256. Otherwise OSR_ERR_UNSUPPORTED with nothing launched: zero dfeat and
 * call osr_roi_align_bwd. out_dtype: OSR_F32, or dout's dtype -- the fp32 sums are then rounded once on the way out (what a separate
 * cast of the fp32 pyramid would give), dfeat->data pointing at tensors of that type. Same reference lines as osr_roi_align_bwd. */
osr_status osr_roi_align_bwd_dense(const osr_pyramid* dfeat, int32_t n, const float* boxes, const int32_t* batch_idx, int64_t m,
                                   int32_t rois_per_image, int32_t pooled, int32_t canonical_level, int32_t canonical_size, int32_t min_level,
                                   const void* dout, int32_t dout_dtype, int32_t out_dtype, void* stream);

/* g[i] = act[i] > 0 ? g[i] : 0, in place (gradient through a ReLU whose output is act). */
osr_status osr_relu_mask(void* g, int32_t g_dtype, const void* act, int32_t act_dtype, int64_t n, void* stream);
/* out[i] = (T)(a_f32[i] + b[i]); either addend may be null. */
osr_status osr_add_cast(const float* a_f32, const void* b, void* out, int32_t dtype, int64_t n, void* stream);
/* mode 0: FPN nearest-2x upsample-add backward, out (n,ho,wo,c) = base + 2x2 sums of src (n,hs,ws,c), ho = ceil(hs/2);
 * mode 1: LastLevelMaxPool (p6 = p5[::2,::2]) backward, out (n,ho,wo,c) = base, plus src[y/2][x/2] at even (y,x). */
osr_status osr_pool_bwd(const void* src, int32_t hs, int32_t ws, const void* base, void* out, int32_t n, int32_t ho,
                        int32_t wo, int32_t c, int32_t mode, int32_t dtype, void* stream);

/* SGD with momentum and weight decay ([d2] build_optimizer -> torch.optim.SGD): g' = grad*grad_scale*row_scale + wd*param;
 * buf = momentum*buf + g'; param -= lr*buf; lowp_copy (nullable) = (T)(param*row_scale). row_scale (nullable): folded
 * FrozenBN scale per leading-dimension row of row_elems elements. apply_flag (nullable, device int32): when it reads 0 the
 * launch changes nothing (the overflow guard of the fp16 step, see osr_check_finite). */
osr_status osr_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                        float weight_decay, float grad_scale, const float* row_scale, int64_t row_elems, void* lowp_copy,
                        int32_t lowp_dtype, const int32_t* apply_flag, void* stream);

/* Backward-data weight of a convolution, (cout,kh,kw,cin) -> (cin,kh,kw,cout) spatially flipped (the forward MFMA kernel run on
 * it with padding k-1-pad is the data gradient; a 1x1 layer / FC matrix is transposed). Run once per SGD step on the refreshed
 * working copies. dtype: element type of both tensors (f16/bf16/f32). out must not alias weight. */
osr_status osr_pack_dgrad_weight(const void* weight, void* out, int32_t cout, int32_t kh, int32_t kw, int32_t cin,
                                 int32_t dtype, void* stream);

/* The update of a whole training step as two launches (csrc/osr_multi_tensor.hip): osr_sgd_step over every entry of a DEVICE-resident
 * table, and osr_pack_dgrad_weight over every entry of another. `chunks`: (tensor index, chunk index) int32 pairs, one per workgroup,
 * on the device: for the SGD launch chunk c of tensor t is the elements [c * chunk_elems, min((c + 1) * chunk_elems, n)); for the packing
 * launch it is the linear index of a 32 x 32 tile, (tap * ceil(cout / 32) + co_tile) * ceil(cin / 32) + ci_tile. The caller builds both
 * once (the buffers' addresses are stable across steps). Same arithmetic per element as the single-tensor entry points: bit-identical. */
typedef struct osr_sgd_tensor {
    float* param;
    const float* grad;
    float* momentum;
    const float* row_scale; /* nullable */
    void* lowp;             /* nullable: low-precision working copy, lowp_dtype */
    int64_t n;
    int64_t row_elems;      /* >= 1 */
    int32_t lowp_dtype;
    int32_t reserved;
} osr_sgd_tensor;
typedef struct osr_pack_tensor {
    const void* src;        /* (cout, kh, kw, cin) */
    void* dst;              /* (cin, kh, kw, cout), spatially flipped */
    int32_t cout, kh, kw, cin;
    int32_t elem_bytes;     /* 2 or 4 */
    int32_t reserved;
} osr_pack_tensor;
osr_status osr_sgd_step_multi(const osr_sgd_tensor* table, const int32_t* chunks, int32_t num_chunks, int32_t chunk_elems,
                              float lr, float momentum, float weight_decay, float grad_scale, const int32_t* apply_flag,
                              void* stream);
osr_status osr_pack_dgrad_weight_multi(const osr_pack_tensor* table, const int32_t* chunks, int32_t num_chunks, void* stream);

/* Overflow guard: *flag (device int32, preset to 1 by the caller) is cleared when any of the n floats of x is inf or NaN.
 * The reference trains in fp32 and has no such step (train.py:135-146); the fp16 gradients of this build do, and an
 * overflowed iteration must not reach the fp32 masters or a checkpoint. x 16-byte aligned. Asynchronous, no host sync. */
osr_status osr_check_finite(const float* x, int64_t n, int32_t* flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OSR_H_ */
