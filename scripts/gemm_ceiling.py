"""Library fp16 GEMM rates on this box (random and all-zero operands): the practical MFMA ceiling
that the planes conv kernels' raw MFMA rate (3 passes per product) is compared with in DESIGN.md."""
import torch, time
dev = "cuda:0"
def rate(M, N, K, zeros, dtype=torch.float16, iters=30):
    a = (torch.zeros if zeros else torch.randn)(M, K, device=dev, dtype=dtype)
    b = (torch.zeros if zeros else torch.randn)(K, N, device=dev, dtype=dtype)
    for _ in range(5): c = a @ b
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9, ms
for (M, N, K) in [(8192, 8192, 8192), (86528, 256, 1152), (86528, 128, 2304), (21632, 512, 2304), (5408, 1024, 4608), (86528, 256, 128)]:
    for z in (0, 1):
        tf, ms = rate(M, N, K, z)
        print(f"M={M} N={N} K={K} zeros={z}: {tf:8.1f} TFLOP/s  {ms*1e3:8.1f} us", flush=True)
