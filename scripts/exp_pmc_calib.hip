// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths RoIAlign uses (MI355X_MICROARCH.md: only 16 B per lane
// is calibrated). Each kernel moves a known byte count through buffers larger than the Infinity Cache:
//   copy16 / copy8      coalesced streaming copy, 16 / 8 bytes per lane
//   gather8             each wave reads 512-byte pixel lines (64 lanes x 8 B) at pseudo-random line indices, no reuse; writes 8 B/lane
//                       coalesced, one 512-byte line per wave-store (the forward RoIAlign pattern)
// Build: hipcc -O3 --offload-arch=gfx950 scripts/exp_pmc_calib.hip -o gpurun_out/calib ; run under rocprofv3 --pmc FETCH_SIZE, then
// --pmc WRITE_SIZE; bytes expected per launch are printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy8(const uint2* __restrict__ in, uint2* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void gather8(const uint2* __restrict__ in, uint2* __restrict__ out, long long lines_in, long long lines_out) {
    const long long w = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;  // wave = output line
    const int lane = threadIdx.x & 63;
    if (w >= lines_out) return;
    unsigned acc0 = 0, acc1 = 0;
    for (int j = 0; j < 8; ++j) {  // 8 input lines per output line
        unsigned long long h = (unsigned long long)(w * 8 + j) * 0x9E3779B97F4A7C15ull;
        const long long line = (long long)((h >> 20) % (unsigned long long)lines_in);
        const uint2 v = in[line * 64 + lane];
        acc0 += v.x; acc1 ^= v.y;
    }
    out[w * 64 + lane] = make_uint2(acc0, acc1);
}
int main() {
    const long long bytes = 1ll << 30;
    void *a, *b;
    hipMalloc(&a, 2 * bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, 2 * bytes); hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        long long n16 = bytes / 16, n8 = bytes / 8;
        hipLaunchKernelGGL(copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16);
        hipLaunchKernelGGL(copy8, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, (const uint2*)a, (uint2*)b, n8);
        const long long lines_in = 2 * bytes / 512, lines_out = bytes / 512 / 8;  // reads 8 x 128 MiB = 1 GiB, writes 128 MiB
        hipLaunchKernelGGL(gather8, dim3((unsigned)((lines_out * 64 + 255) / 256)), dim3(256), 0, 0, (const uint2*)a, (uint2*)b, lines_in, lines_out);
        hipDeviceSynchronize();
    }
    printf("expected per launch: copy16 read %lld write %lld | copy8 read %lld write %lld | gather8 read %lld write %lld\n", bytes, bytes, bytes, bytes,
           bytes, bytes / 8);
    return 0;
}
