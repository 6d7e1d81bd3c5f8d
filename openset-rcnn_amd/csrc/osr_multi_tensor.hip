// The parameter update of a training step as TWO launches instead of ~145: torch.optim.SGD.step over every parameter tensor
// ([d2] build_optimizer; /root/reference/train.py:146 `optimizer.step()`), and the repacking of every convolution / FC weight into
// its backward-data layout. One launch per tensor is a few microseconds of work behind a launch each -- 0.48 ms of a 24.7 ms
// iteration against a bandwidth floor of 0.2 ms, and 145 host-side launches per iteration and rank.
// The caller builds two tables once (pointers of the trainer's buffers are stable), uploads them, and passes them every step:
//   table: one entry per tensor; chunks: (tensor index, chunk index) per workgroup, chunk = a fixed run of elements / one 32 x 32 tile.
// Per-element arithmetic and its order are those of osr_sgd_step / osr_pack_dgrad_weight: the results are bit-identical (tested).
#include "osr_common.h"

// ---- SGD ---------------------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void sgd_run(const osr_sgd_tensor& t, long long i0, long long i1, float lr, float mu, float wd, float gs) {
    float* __restrict__ p = t.param;
    const float* __restrict__ g = t.grad;
    float* __restrict__ v = t.momentum;
    T* __restrict__ lp = reinterpret_cast<T*>(t.lowp);
    for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float rs = t.row_scale ? t.row_scale[i / t.row_elems] : 1.0f;
        float pi = p[i], vi = v[i];
        osr_sgd_element(pi, vi, g[i], rs, lr, mu, wd, gs);
        v[i] = vi;
        p[i] = pi;
        if (lp) lp[i] = osr_from_float<T>(pi * rs);
    }
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(const osr_sgd_tensor* __restrict__ table, const int2* __restrict__ chunks, int chunk_elems, float lr,
                                                        float mu, float wd, float gs, const int* __restrict__ gate) {
    if (gate && *gate == 0) return;  // this iteration's gradients held an inf / NaN (osr_check_finite): leave parameters and momentum alone
    const int2 c = chunks[blockIdx.x];
    const osr_sgd_tensor t = table[c.x];
    const long long i0 = (long long)c.y * chunk_elems, i1 = i0 + chunk_elems < t.n ? i0 + chunk_elems : t.n;
    if (!t.lowp || t.lowp_dtype == OSR_F32) sgd_run<float>(t, i0, i1, lr, mu, wd, gs);
    else if (t.lowp_dtype == OSR_F16) sgd_run<f16_t>(t, i0, i1, lr, mu, wd, gs);
    else sgd_run<bf16_t>(t, i0, i1, lr, mu, wd, gs);
}

extern "C" osr_status osr_sgd_step_multi(const osr_sgd_tensor* table, const int32_t* chunks, int32_t num_chunks, int32_t chunk_elems, float lr, float momentum,
                                         float weight_decay, float grad_scale, const int32_t* apply_flag, void* stream) {
    OSR_REQUIRE(table && chunks && num_chunks >= 0 && chunk_elems >= 256, OSR_ERR_INVALID_ARG, "osr_sgd_step_multi: null table / bad chunk size");
    if (num_chunks == 0) return OSR_OK;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)num_chunks), dim3(256), 0, (hipStream_t)stream, table, reinterpret_cast<const int2*>(chunks), chunk_elems, lr,
                       momentum, weight_decay, grad_scale, apply_flag);
    OSR_CHECK_LAUNCH("osr_sgd_step_multi");
    return OSR_OK;
}

// ---- backward-data weights -----------------------------------------------------------------------------------------------------
// dst[ci][kh-1-y][kw-1-x][co] = src[co][y][x][ci]; a chunk = one 32 x 32 (co, ci) tile of one tap, through LDS, both sides coalesced.
template <class T>
__device__ __forceinline__ void pack_tile(const osr_pack_tensor& t, int tile, unsigned char* smem) {
    T (*buf)[33] = reinterpret_cast<T (*)[33]>(smem);
    const int tiles_ci = (t.cin + 31) >> 5, tiles_co = (t.cout + 31) >> 5;
    const int bx = tile % tiles_ci, rest = tile / tiles_ci, by = rest % tiles_co, tap = rest / tiles_co;
    const int y = tap / t.kw, x = tap % t.kw;
    const int ci0 = bx * 32, co0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const long long taps = (long long)t.kh * t.kw;
    const T* __restrict__ src = reinterpret_cast<const T*>(t.src);
    T* __restrict__ dst = reinterpret_cast<T*>(t.dst);
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < t.cout && ci < t.cin) buf[r][tx] = src[((long long)co * taps + tap) * t.cin + ci];
    }
    __syncthreads();
    const int ftap = (t.kh - 1 - y) * t.kw + (t.kw - 1 - x);
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < t.cin && co < t.cout) dst[((long long)ci * taps + ftap) * t.cout + co] = buf[tx][r];
    }
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const osr_pack_tensor* __restrict__ table, const int2* __restrict__ chunks) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[32 * 33 * 4];
    const int2 c = chunks[blockIdx.x];
    const osr_pack_tensor t = table[c.x];
    if (t.elem_bytes == 4) pack_tile<float>(t, c.y, smem);
    else pack_tile<unsigned short>(t, c.y, smem);
}

extern "C" osr_status osr_pack_dgrad_weight_multi(const osr_pack_tensor* table, const int32_t* chunks, int32_t num_chunks, void* stream) {
    OSR_REQUIRE(table && chunks && num_chunks >= 0, OSR_ERR_INVALID_ARG, "osr_pack_dgrad_weight_multi: null table");
    if (num_chunks == 0) return OSR_OK;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)num_chunks), dim3(256), 0, (hipStream_t)stream, table, reinterpret_cast<const int2*>(chunks));
    OSR_CHECK_LAUNCH("osr_pack_dgrad_weight_multi");
    return OSR_OK;
}
