import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
M, N, K, tile, st = [int(x) for x in sys.argv[1:6]]
X = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(10): ops.gemm(X, W, out, tile=tile, splitk=1, stages=st)
torch.cuda.synchronize()
