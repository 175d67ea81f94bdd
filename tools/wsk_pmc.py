"""Eager launches of the 1024 x 1280 x K product on the wave-split-K and the tiled kernel, for `rocprofv3 --pmc ...` (tools/wsk_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import ops as O
BF = torch.bfloat16
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
M, N = 1024, 1280
x = torch.randn(M, K, device="cuda").to(BF)
ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(BF) for _ in range(8)]
y = torch.zeros(M, N, device="cuda", dtype=BF)
for i in range(24):
    O.WSK = True
    O.gemm(x, ws[i % 8], y)
    O.WSK = False
    O.gemm(x, ws[i % 8], y)
torch.cuda.synchronize()
