"""Scratch: a few launches of the self-attention forward / backward at one SDXL shape, for a rocprofv3 --pmc pass (tools/attn32_pmc.sh)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16
N, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 10)
B, d = 1, 64
C = H * d
r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
Q, K, V, dO = r(N), r(N), r(N), r(N)
O = torch.zeros(B * N, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * N, device="cuda"); D = torch.zeros_like(L)
dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
kw = dict(B=B, H=H, Nq=N, Nk=N, Nqp=N, Nkp=N, d=d, scale=1 / math.sqrt(d))
for _ in range(8):
    ops.attn_fwd(Q, K, V, None, O, L, **kw)
    ops.attn_bwd(Q, K, V, None, None, O, L, dO, None, D, dQ, dK, dV, **kw)
torch.cuda.synchronize()
