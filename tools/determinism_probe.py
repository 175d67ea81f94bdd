"""Scratch: run kernels / the whole step twice from identical state and compare bitwise (race detector)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
from oracle import unet_ref as U
import sd_lora_trainer_amd.step as S, sd_lora_trainer_amd.unet as M
from sd_lora_trainer_amd import topology
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(BF).cuda()

def gemm_case(M_, N, K, r, tile, splitk, conv=None):
    X = rnd(M_, K) if conv is None else rnd(conv.B * conv.Hin * conv.Win, conv.Cin)
    W = rnd(N, K, sc=1 / math.sqrt(K))
    lora = None
    if r:
        lora = (rnd(16, K, sc=0.05), rnd(N, 16, sc=0.3), 1.0, torch.empty(M_, 16, dtype=BF, device="cuda"))
    outs = []
    for i in range(6):
        o = torch.empty(M_, N, dtype=BF, device="cuda")
        ops.gemm(X, W, o, lora=lora, conv=conv, tile=tile, splitk=splitk)
        outs.append(o.clone())
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
    md = max(float((outs[0].float() - o.float()).abs().max()) for o in outs[1:])
    print(f"gemm M{M_} N{N} K{K} r{r} tile{tile} splitk{splitk} conv{conv is not None}: {bad}/5 differ, maxdiff {md:.4g}")

for tile in (1, 2, 3):
    gemm_case(1024, 1280, 1280, 0, tile, 1)
    gemm_case(1024, 1280, 1280, 16, tile, 1)
    gemm_case(4096, 640, 5120, 0, tile, 1)
gemm_case(1024, 1280, 1280, 16, 0, 0)
gemm_case(1024, 1280, 5120, 0, 0, 0)
gemm_case(1024, 320, 9 * 320, 0, 0, 1, conv=ops.ConvGeom(1, 32, 32, 320, 32, 32))
gemm_case(1024, 320, 9 * 320, 16, 0, 0, conv=ops.ConvGeom(1, 32, 32, 320, 32, 32))

# whole step determinism, eager
for version, B in (("tiny15", 2), ("tinyxl", 1)):
    cfg = U.CONFIGS[version]; h = 16
    sd = U.init_unet_state(cfg, seed=0); lora = U.init_lora(cfg, 4, seed=1, b_std=0.05)
    rt = M.Runtime("cuda:0", B); unet = M.UNet(rt, topology.CONFIGS[version], sd, lora_rank=4); unet.arena.load(lora)
    ts = S.TrainStep(rt, unet, latent_hw=(h, h))
    gg = torch.Generator().manual_seed(3)
    lat = torch.randn(B, 4, h, h, generator=gg); noi = torch.randn(B, 4, h, h, generator=gg); msk = torch.ones(B, 4, h, h)
    t = torch.tensor([10, 900][:B]); ctx = torch.randn(B, 77, cfg["cross_dim"], generator=gg)
    pooled = tid = None
    if cfg["addition"]:
        pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=gg).cuda(); tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B).cuda()
    ts.set_batch(lat.cuda(), noi.cuda(), t.cuda(), msk.cuda(), ctx.cuda(), pooled, tid)
    res = []
    for i in range(4):
        pred = ts.forward_backward().clone(); torch.cuda.synchronize()
        res.append((pred, unet.arena.grads.clone(), ts.dctx.clone()))
    for i in range(1, 4):
        dp = float((res[0][0] - res[i][0]).abs().max()); dg = float((res[0][1] - res[i][1]).norm() / res[0][1].norm()); dc = float((res[0][2].float() - res[i][2].float()).norm() / res[0][2].float().norm())
        print(f"{version} run0 vs run{i}: pred maxdiff {dp:.3g}, grads rel {dg:.3g}, dctx rel {dc:.3g}")
