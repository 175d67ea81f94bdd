import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_ref as U
import sd_lora_trainer_amd.step as S, sd_lora_trainer_amd.unet as M
from sd_lora_trainer_amd import topology
version, B, rank, h = "tinyxl", 1, 16, 16
cfg = U.CONFIGS[version]
sd = U.init_unet_state(cfg, seed=0); lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
rt = M.Runtime("cuda:0", B); unet = M.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank); unet.arena.load(lora)
ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004)
gg = torch.Generator().manual_seed(3)
lat = torch.randn(B, 4, h, h, generator=gg) * 0.13; noi = torch.randn(B, 4, h, h, generator=gg); msk = torch.ones(B, 4, h, h)
t = torch.tensor([10, 900][:B]); ctx = torch.randn(B, 77, cfg["cross_dim"], generator=gg)
pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=gg).cuda(); tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B).cuda()
ts.set_batch(lat.cuda(), noi.cuda(), t.cuda(), msk.cuda(), ctx.cuda(), pooled, tid)
a = unet.arena
def st(tag):
    torch.cuda.synchronize()
    f = lambda x: f"{float(x.abs().max()):.3g}/{int((~torch.isfinite(x)).sum())}"
    print(tag, "loss", float(ts.loss), "params", f(a.params), "grads", f(a.grads), "m", f(a.m), "v", f(a.v), "dctx", f(ts.dctx.float()))
    bad = [e["name"] for e in a.entries if not torch.isfinite(e["gA"]).all() or not torch.isfinite(e["gB"]).all()]
    if bad: print("   first bad grads:", bad[:6], len(bad))
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
ts.forward_backward(); st("eager fwd/bwd")
if "hyper" in mode:
    ts.set_hyper(1e-3); ts.opt_step = 0
if "fbonly" in mode:
    ts.body = ts.forward_backward
if "nosplit" in mode:
    from sd_lora_trainer_amd import ops as O
    _g = O.gemm
    def g2(*a, **k):
        k["splitk"] = 1
        return _g(*a, **k)
    O.gemm = g2
p0 = a.params.clone()
ts.capture(warmup=1); st("after capture")
print("params restored exactly:", bool(torch.equal(a.params, p0)))
def shadow_err():
    worst = 0.0
    for e in a.entries:
        r = a.rank
        worst = max(worst, float((e["A_s"][:r].float() - e["A"].to(torch.bfloat16).float()).abs().max()), float((e["B_s"][:, :r].float() - e["B"].to(torch.bfloat16).float()).abs().max()))
    return worst
print("shadow err after capture:", shadow_err())
ts.forward_backward(); st("eager after capture")
for i in range(3):
    ts.run(1e-3)
    if "fbonly" in mode: ts.optimizer_step()
    st(f"replay {i}")
