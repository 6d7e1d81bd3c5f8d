// does the SHAPE of the pieces alone explain why the 1x1 layers of res3-res5 move 3.3-4.2 TB/s where a
// streaming copy moves 6.0-6.3? (docs/HISTORY.md, round 3: "every workgroup reads and writes 256-byte pieces of 2-KB pixel rows, the
// eight N tiles of a row block complete it at different times, so DRAM sees line-sized accesses with little page locality" -- never tested.)
//
// No product code: plain loads / stores of a (rows x 2048 B) activation matrix (res4: 67 200 pixel rows x 1024 fp16 channels) in the
// shapes the conv kernel's tiles produce, against whole-row shapes moving the same bytes with the same number of workgroups, threads,
// bytes per workgroup, bytes in flight per step and workgroups per CU (3, capped through the LDS allocation like the 128 x 128 kernel):
//
//   write  piece256   workgroup (mt, nt) stores rows [128 mt, +128) x bytes [256 nt, +256)  = conv3's epilogue (N tile = 128 channels);
//                     tile order as the kernel's: XCD-aware, the 8 N tiles of a row block adjacent on one XCD
//   write  piece256s  the same pieces, N-major order (all row blocks of nt = 0 first ...): the eight pieces of a row land far apart in time
//   write  row        workgroup t stores rows [16 t, +16) whole (32 KB contiguous)
//   write  rowtile64  workgroup t stores rows [64 t, +64) whole (128 KB contiguous: VERDICT's "row-complete tile"), 4x fewer workgroups
//   read   piece256 / piece256s / row       the residual prefetch of conv3, same shapes
//   read   kslice128  workgroup mt reads rows [128 mt, +128) x bytes [128 k, +128) for k = 0..15, one slice per step with a full wait
//                     between steps = conv1's A operand (K = 1024, BK = 64) in the single-buffer loop; only tiles_n = 1 (cout 256 / BN 128
//                     is 2: the second N tile re-reads from L2), so M/128 workgroups x 16 dependent steps of 16 KB
//   read   kslice512  the same rows as 4 steps of 128 rows x 512 B (BK = 256)
//   read   rowstep    rows [128 mt, +128) as 16 dependent steps of 16 KB CONTIGUOUS (8 whole rows per step)
//
// Every launch works on the next of NBUF buffers (1.1 GB in rotation > the 256 MB Infinity Cache): cold like a layer's activations.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/exp_piece_shape.hip -o gpurun_out/exp_piece_shape ; run: gpurun_out/exp_piece_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int ROWB = 2048;      // bytes per pixel row (1024 fp16 channels)
constexpr int M = 67200;        // 16 x 50 x 84 (res4 at batch 16); 525 row blocks of 128
constexpr int NBUF = 8;

__device__ __forceinline__ int xcd_tile(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0: piece256 (XCD-aware, N tiles adjacent), 1: piece256s (N-major), 2: row (16 whole rows), 3: rowtile64 (64 whole rows)
template <bool WRITE>
__global__ __launch_bounds__(256) void shape_kernel(unsigned char* __restrict__ buf, int mode, int rows, u32x4* __restrict__ sink) {
    extern __shared__ unsigned char lds_cap[];  // occupancy cap only
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, gridDim.x);
    u32x4 acc = {0u, 0u, 0u, 0u};
    const u32x4 val = {(unsigned)t, (unsigned)tid, 0x3c003c00u, 0x3c003c00u};
    if (mode <= 1) {
        const int tiles_m = (rows + 127) / 128;
        const int mt = mode == 0 ? t / 8 : t % tiles_m, nt = mode == 0 ? t % 8 : t / tiles_m;
        // a wave-instruction = 4 rows x 256 B (16 lanes x 16 B per row); 256 threads = 16 rows per step, 8 steps
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = mt * 128 + s * 16 + (tid >> 4);
            if (row < rows) {
                u32x4* p = reinterpret_cast<u32x4*>(buf + (size_t)row * ROWB + nt * 256 + (tid & 15) * 16);
                if (WRITE) *p = val; else { const u32x4 v = *p; acc ^= v; }
            }
        }
    } else {
        const int nrow = mode == 2 ? 16 : 64;
        // contiguous: 256 threads x 16 B = 4 KB (two rows) per step
        for (int s = 0; s < nrow / 2; ++s) {
            const size_t off = ((size_t)t * nrow + s * 2) * ROWB + (size_t)tid * 16;
            if (off < (size_t)rows * ROWB) {
                u32x4* p = reinterpret_cast<u32x4*>(buf + off);
                if (WRITE) *p = val; else { const u32x4 v = *p; acc ^= v; }
            }
        }
    }
    if (!WRITE && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;  // (never true: keeps the loads)
}

// dependent-step reads of a 128-row block: mode 0 kslice128 (16 steps x 128 rows x 128 B), 1 kslice512 (4 steps x 128 rows x 512 B),
// 2 rowstep (16 steps x 16 KB contiguous), 3 kslice128 with TWO slices in flight (double buffer)
__global__ __launch_bounds__(256) void kslice_kernel(const unsigned char* __restrict__ buf, int mode, int rows, u32x4* __restrict__ sink) {
    extern __shared__ unsigned char lds_cap[];
    const int tid = threadIdx.x;
    const int mt = xcd_tile(blockIdx.x, gridDim.x);
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (mode == 0 || mode == 3) {
        // per step: 128 rows x 128 B = 8 lanes per row, 32 rows per pass of 256 threads, 4 loads per thread
        u32x4 prev[4];
        for (int k = 0; k < 16; ++k) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = min(mt * 128 + j * 32 + (tid >> 3), rows - 1);
                v[j] = *reinterpret_cast<const u32x4*>(buf + (size_t)row * ROWB + k * 128 + (tid & 7) * 16);
            }
            if (mode == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) acc ^= v[j];
            } else {
                if (k > 0) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc ^= prev[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) prev[j] = v[j];
            }
        }
        if (mode == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= prev[j];
        }
    } else if (mode == 1) {
        for (int k = 0; k < 4; ++k) {
            u32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {  // 32 lanes per row, 8 rows per pass
                const int row = min(mt * 128 + j * 8 + (tid >> 5), rows - 1);
                v[j] = *reinterpret_cast<const u32x4*>(buf + (size_t)row * ROWB + k * 512 + (tid & 31) * 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= v[j];
        }
    } else {
        for (int k = 0; k < 16; ++k) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                size_t off = ((size_t)mt * 128 + k * 8) * ROWB + (size_t)j * 4096 + (size_t)tid * 16;
                off = min(off, (size_t)rows * ROWB - 16);
                v[j] = *reinterpret_cast<const u32x4*>(buf + off);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= v[j];
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

// conv3-like: per workgroup (mt, nt): read the residual piece (128 rows x 256 B), then write the output piece -- or both as whole rows
__global__ __launch_bounds__(256) void rw_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int mode, int rows) {
    extern __shared__ unsigned char lds_cap[];
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, gridDim.x);
    if (mode == 0) {
        const int mt = t / 8, nt = t % 8;
        u32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = min(mt * 128 + s * 16 + (tid >> 4), rows - 1);
            v[s] = *reinterpret_cast<const u32x4*>(src + (size_t)row * ROWB + nt * 256 + (tid & 15) * 16);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = mt * 128 + s * 16 + (tid >> 4);
            if (row < rows) *reinterpret_cast<u32x4*>(dst + (size_t)row * ROWB + nt * 256 + (tid & 15) * 16) = v[s] + 1u;
        }
    } else {
        u32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const size_t off = min(((size_t)t * 16 + s * 2) * ROWB + (size_t)tid * 16, (size_t)rows * ROWB - 16);
            v[s] = *reinterpret_cast<const u32x4*>(src + off);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const size_t off = ((size_t)t * 16 + s * 2) * ROWB + (size_t)tid * 16;
            if (off < (size_t)rows * ROWB) *reinterpret_cast<u32x4*>(dst + off) = v[s] + 1u;
        }
    }
}

template <class F>
static double time_ms(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch(i);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / iters);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)M * ROWB;
    unsigned char* bufs[NBUF];
    for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&bufs[i], bytes)); CK(hipMemset(bufs[i], i + 1, bytes)); }
    u32x4* sink;
    CK(hipMalloc(&sink, 64));
    const int lds_cap = 50 * 1024;  // 3 workgroups per CU, like the 128 x 128 single-buffer kernel
    const int iters = 16;
    const int tiles_m = (M + 127) / 128;
    printf("rows %d x %d B = %.1f MB per buffer, %d buffers in rotation, %d-B LDS cap (3 workgroups per CU), median of 5 x %d launches\n", M, ROWB, bytes / 1e6, NBUF,
           lds_cap, iters);
    struct Row { const char* name; double ms; double gb; };
    std::vector<Row> out;
    const char* wnames[4] = {"piece256 (kernel's tile order)", "piece256s (N-major: pieces of a row far apart in time)", "row (16 whole rows per workgroup)",
                             "rowtile64 (64 whole rows per workgroup)"};
    for (int wr = 1; wr >= 0; --wr)
        for (int mode = 0; mode < 4; ++mode) {
            const int grid = mode <= 1 ? tiles_m * 8 : mode == 2 ? (M + 15) / 16 : (M + 63) / 64;
            const double ms = time_ms([&](int i) {
                if (wr) hipLaunchKernelGGL(shape_kernel<true>, dim3(grid), dim3(256), lds_cap, 0, bufs[i % NBUF], mode, M, sink);
                else hipLaunchKernelGGL(shape_kernel<false>, dim3(grid), dim3(256), lds_cap, 0, bufs[i % NBUF], mode, M, sink);
            }, iters);
            char* nm = (char*)malloc(160);
            snprintf(nm, 160, "%s %s", wr ? "write" : "read ", wnames[mode]);
            out.push_back({nm, ms, bytes / 1e9});
        }
    const char* knames[4] = {"read  kslice128 (16 dependent steps of 128 rows x 128 B)", "read  kslice512 (4 dependent steps of 128 rows x 512 B)",
                             "read  rowstep (16 dependent steps of 16 KB contiguous)", "read  kslice128, two slices in flight"};
    for (int mode = 0; mode < 4; ++mode) {
        const double ms = time_ms([&](int i) { hipLaunchKernelGGL(kslice_kernel, dim3(tiles_m), dim3(256), lds_cap, 0, bufs[i % NBUF], mode, M, sink); }, iters);
        out.push_back({knames[mode], ms, bytes / 1e9});
    }
    const char* rwnames[2] = {"read + write piece256 (conv3: residual in, output out)", "read + write whole rows"};
    for (int mode = 0; mode < 2; ++mode) {
        const int grid = mode == 0 ? tiles_m * 8 : (M + 15) / 16;
        const double ms = time_ms([&](int i) { hipLaunchKernelGGL(rw_kernel, dim3(grid), dim3(256), lds_cap, 0, bufs[i % NBUF], bufs[(i + NBUF / 2) % NBUF], mode, M); }, iters);
        out.push_back({rwnames[mode], ms, 2 * bytes / 1e9});
    }
    CK(hipDeviceSynchronize());
    printf("%-64s %9s %9s\n", "shape", "us", "TB/s");
    for (auto& r : out) printf("%-64s %9.1f %9.2f\n", r.name, r.ms * 1e3, r.gb / r.ms);
    return 0;
}
