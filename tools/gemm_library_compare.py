#!/usr/bin/env python3
"""What the vendor GEMM library reaches on this box at the shapes of ECAPA's wide layers -- a ceiling proxy for k_conv_gemm_g256 (fp16) and
k_conv_gemm_w256 (f32), whose measured rates stand beside it in profiles/.  Measurement only: nothing in the product path uses a BLAS library
(libsdhip.so links HIP + RCCL only); torch.matmul is just the handiest way to reach hipBLASLt / rocBLAS from this image.
    python tools/gemm_library_compare.py > gpurun_out/gemm_library_compare.txt
Shapes: rows = frames of one embedding batch of the planted hour (3 072 items x ~280 live frames), [rows, K] x [K, N]:
tdnn1 / tdnn2 1024 -> 1024, MFA 3072 -> 3072, plain row-major activations and [N, K] weights as in the package (y = x @ w.T)."""
import time, torch

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), "torch", torch.__version__)
torch.backends.cuda.matmul.allow_tf32 = False


def run(M, K, N, dtype, iters=20):
    x = torch.randn(M, K, device=dev, dtype=torch.float32).to(dtype)
    w = torch.randn(N, K, device=dev, dtype=torch.float32).to(dtype) * 0.03
    y = torch.empty(M, N, device=dev, dtype=dtype)
    for _ in range(3):
        torch.matmul(x, w.t(), out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(x, w.t(), out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * K * N / ms / 1e9


for name, K, N in (("tdnn 1024 -> 1024", 1024, 1024), ("MFA 3072 -> 3072", 3072, 3072)):
    for M in (262144, 860160, 215040):
        for dtype, peak in ((torch.float16, 2500.0), (torch.float32, 157.3)):
            try:
                ms, tf = run(M, K, N, dtype)
                print(f"{name:20s} rows {M:7d} {str(dtype):14s} {ms:9.3f} ms {tf:8.1f} TFLOP/s  {100 * tf / peak:5.1f} % of {peak:g}")
            except Exception as ex:                      # noqa
                print(name, M, dtype, "failed:", str(ex)[:200])
