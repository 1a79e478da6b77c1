"""Which kernels does the library GEMM (hipBLASLt / rocBLAS behind torch.matmul) launch for the Stage-2 transformer shapes?
Run under `rocprofv3 --kernel-trace --stats`: the Tensile kernel names spell out macro tile, MFMA shape, LDS / prefetch
options and workgroup size -- a calibration of what the hardware allows on these shapes (not used by the product)."""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, K, N) in [(32768, 1280, 1280), (32768, 1280, 3840), (32768, 1280, 10240), (32768, 5120, 1280), (131072, 640, 1920), (131072, 2560, 640)]:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(5):
        y = x @ w.t()
    torch.cuda.synchronize()
    print(M, K, N, "done", flush=True)
