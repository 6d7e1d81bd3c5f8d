// Experiment (round 6, RoIAlign): what does a buffer load cost the CU's vector-memory path when every lane is out of the buffer's range
// (voffset >= num_records: returns zeros, touches no cache line)? The streaming RoIAlign kernel issues D groups of prefetches past the end
// of every stream; today they re-read the last column (L1 hits). If an out-of-range load is much cheaper than an L1 hit, pointing the dead
// prefetches out of range removes their cost without touching the loop's structure.
// Every wave streams a 16-KiB (L1-resident) window with 16-byte buffer loads; mode 0: all in range, 1: all out of range,
// 2: every second instruction out of range, 3: all lanes' EXEC off for every second instruction (lane-predicated).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void rd(const char* __restrict__ p, unsigned* out, int win_bytes, int iters, int dead) {
    const char* base = p + (size_t)blockIdx.x * win_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, win_bytes, 0x00020000);
    const int lane_off = threadIdx.x * 16;
    const int voff_dead = lane_off + dead;  // dead = 0x7ff00000: out of range
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int o = 0; o < win_bytes; o += 256 * 16) {
            int vo = lane_off;
            if (MODE == 1) vo = voff_dead;
            if (MODE == 2) vo = (o & 4096) ? voff_dead : lane_off;
            u4 v;
            if (MODE == 3) {
                if ((o & 4096) == 0 || threadIdx.x + dead == 77) v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, o, 0); else v = u4{0u, 0u, 0u, 0u};
            } else v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, o, 0);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    char* p; unsigned* o;
    const int nwg = 1024, win = 16 << 10;
    hipMalloc(&p, (size_t)nwg * win); hipMalloc(&o, 4);
    hipMemset(p, 1, (size_t)nwg * win);
    const char* names[4] = {"all in range (L1 hits)", "all out of range", "every second out of range", "every second skipped by a (divergent-looking) branch"};
    for (int mode = 0; mode < 4; ++mode) {
        const int iters = (64 << 20) / win;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(rd<0>, dim3(nwg), dim3(256), 0, 0, p, o, win, iters, 0x7ff00000); break;
                case 1: hipLaunchKernelGGL(rd<1>, dim3(nwg), dim3(256), 0, 0, p, o, win, iters, 0x7ff00000); break;
                case 2: hipLaunchKernelGGL(rd<2>, dim3(nwg), dim3(256), 0, 0, p, o, win, iters, 0x7ff00000); break;
                default: hipLaunchKernelGGL(rd<3>, dim3(nwg), dim3(256), 0, 0, p, o, win, iters, 0x7ff00000); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double winstr = (double)nwg * 4 * ((64 << 20) / 1024 / 4);  // wave-instructions issued (slots), 1 KiB each
        printf("%-56s %8.3f ms  %6.2f G wave-instruction slots/s  (%.1f TB/s if every slot carried 1 KiB)\n", names[mode], ms, winstr / ms / 1e6, winstr * 1024 / ms / 1e9);
    }
    return 0;
}
