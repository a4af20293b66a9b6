"""Host cost of one eager operator call (developer script)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from vllm_quick_amd._lib import lib
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N, G = 4096, 1376, 128
p = ops.awq_repack(*make_weights(K, N, G, 0, dev))
x = make_x(128, K, 1, dev)
for name, fn in (("ops.awq_gemm_packed", lambda: ops.awq_gemm_packed(x, p)),):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {1e6*(t1-t0)/2000:.1f} us/call, with drain {1e6*(t2-t0)/2000:.1f} us/call")
out = torch.empty((128, N), dtype=torch.float16, device=dev)
s = torch.cuda.current_stream().cuda_stream
args = (x.data_ptr(), p.packed_w.data_ptr(), p.packed_sz.data_ptr(), None, out.data_ptr(), 128, K, N, G, s)
for _ in range(200): lib.w4a16_gemm_packed_f16(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): lib.w4a16_gemm_packed_f16(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"C ABI direct: host {1e6*(t1-t0)/2000:.1f} us/call, with drain {1e6*(t2-t0)/2000:.1f} us/call")
