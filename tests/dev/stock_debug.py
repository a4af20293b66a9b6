"""Developer aid: probes of the stock-layout GEMM kernel (which scale each output column gets)."""
import numpy as np
import torch
from vllm_quick_amd._C import ops

dev = torch.device("cuda:0")
K, N, G = 32, 128, 32
qw = torch.full((K, N // 8), 0x11111111, dtype=torch.int32, device=dev)
qz = torch.zeros((K // G, N // 8), dtype=torch.int32, device=dev)
s = torch.arange(1, N + 1, dtype=torch.float16, device=dev).reshape(1, N)
x = torch.zeros((1, K), dtype=torch.float16, device=dev)
x[0, 0] = 1
y = ops.awq_gemm_stock(x, qw, s, qz).cpu().numpy()
print(y[0, :32].tolist())
