// Experiment: bytes per second a CU's vector memory path delivers from cache-resident data with 8 B per lane (dwordx2) against
// 16 B per lane (dwordx4) loads. Every wave streams its workgroup's window (L1-sized or L2-sized) over and over; 1024 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int W> __global__ __launch_bounds__(256) void rd(const char* __restrict__ p, unsigned* out, int win_bytes, int iters) {
    const char* base = p + (size_t)blockIdx.x * win_bytes;
    const int lane_off = threadIdx.x * W;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int o = 0; o < win_bytes; o += 256 * W) {
            if (W == 8) { u2 v = *reinterpret_cast<const u2*>(base + o + lane_off); acc += v.x ^ v.y; }
            else { u4 v = *reinterpret_cast<const u4*>(base + o + lane_off); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    char* p; unsigned* o;
    const int nwg = 1024;
    hipMalloc(&p, (size_t)nwg * (1 << 20)); hipMalloc(&o, 4);
    hipMemset(p, 1, (size_t)nwg * (1 << 20));
    for (int win : {16 << 10, 256 << 10}) {
        for (int w : {8, 16}) {
            const int iters = (64 << 20) / win;  // 64 MiB per workgroup
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (w == 8) hipLaunchKernelGGL(rd<8>, dim3(nwg), dim3(256), 0, 0, p, o, win, iters);
                else hipLaunchKernelGGL(rd<16>, dim3(nwg), dim3(256), 0, 0, p, o, win, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("window %4d KiB, %2d B/lane: %.1f TB/s\n", win >> 10, w, (double)nwg * 64 * (1 << 20) / ms / 1e9);
        }
    }
    return 0;
}
