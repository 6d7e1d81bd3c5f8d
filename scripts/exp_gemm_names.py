"""Which hipBLASLt kernels torch.matmul picks on the product's GEMM shapes (run under rocprofv3 --kernel-trace --stats; the kernel names carry
the macro tile, the MFMA shape and the staging scheme)."""
import math, torch
g = torch.Generator().manual_seed(0)
for m, k, n in [(68368, 12544, 1024), (16 * 200 * 336, 2304, 256), (16800, 4608, 512), (67200, 1024, 256), (8192, 8192, 8192)]:
    a = (torch.randn(m, k, generator=g) * 0.5).half().cuda()
    b = (torch.randn(n, k, generator=g) / math.sqrt(k)).half().cuda()
    out = torch.empty(m, n, dtype=torch.float16, device="cuda")
    for _ in range(6):
        torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()
    del a, b, out
