#include <hip/hip_runtime.h>
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(const short* in, short* out) {
    __shared__ short t[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) t[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    typedef __attribute__((address_space(3))) s4v lds_s4;
    s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(t + (g * 8 + q) * 128 + 4 * p));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short h[64 * 128], o[256];
    for (int i = 0; i < 64 * 128; ++i) h[i] = (short)((i / 128) * 100 + (i % 128));  // row*100 + col
    short *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) { if (l < 20 || l % 16 == 0) printf("lane %2d: %d %d %d %d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]); }
    return 0;
}
