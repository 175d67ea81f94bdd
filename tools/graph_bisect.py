import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_ref as U
import sd_lora_trainer_amd.step as S, sd_lora_trainer_amd.unet as M
from sd_lora_trainer_amd import topology
version, B, rank, h = sys.argv[1] if len(sys.argv) > 1 else "tinyxl", 1, 16, 16
cfg = U.CONFIGS[version]
sd = U.init_unet_state(cfg, seed=0); lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
rt = M.Runtime("cuda:0", B); unet = M.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank); unet.arena.load(lora)
ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004)
gg = torch.Generator().manual_seed(3)
lat = torch.randn(B, 4, h, h, generator=gg) * 0.13; noi = torch.randn(B, 4, h, h, generator=gg); msk = torch.ones(B, 4, h, h)
t = torch.tensor([10, 900][:B]); ctx = torch.randn(B, 77, cfg["cross_dim"], generator=gg)
pooled = tid = None
if cfg["addition"]:
    pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=gg).cuda(); tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B).cuda()
ts.set_batch(lat.cuda(), noi.cuda(), t.cuda(), msk.cuda(), ctx.cuda(), pooled, tid)

def modules(root):
    seen, out = set(), []
    def walk(o, path):
        if id(o) in seen: return
        seen.add(id(o))
        if isinstance(o, M._Module):
            out.append((path, o))
            for k, v in vars(o).items():
                if k in ("rt",): continue
                walk(v, path + "." + k)
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o): walk(v, f"{path}[{i}]")
    walk(root, "unet")
    return out
def snapshot():
    torch.cuda.synchronize()
    d = {}
    for path, m in modules(unet):
        for k, v in m._b.items():
            d[f"{path}:{k}"] = v.detach().float().clone()
    for k in ("x64", "dpred64", "dctx", "loss"):
        d["ts:" + k] = getattr(ts, k).detach().float().clone()
    d["arena:grads"] = unet.arena.grads.clone()
    return d
ts.set_hyper(0.0); ts.opt_step = 0     # lr = 0: parameters never change, every pass must reproduce the same buffers
ts.forward_backward(); ref = snapshot()
ts.forward_backward(); again = snapshot()
ts.capture(warmup=1)
ts.run(0.0); rep = snapshot()
ts.run(0.0); rep2 = snapshot()
ts.run(0.0); rep3 = snapshot()
def cmp(a, b, tag):
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if x.shape != y.shape: bad.append((k, "shape")); continue
        fin = torch.isfinite(y).all()
        sc = float(x.abs().max()) + 1e-12
        err = float((x - y).abs().max()) / sc if fin else float("inf")
        if err > 0.2: bad.append((k, f"{err:.3g}"))
    print(tag, len(bad), "of", len(a), "buffers differ >20%")
    for k, e in bad[:25]: print("   ", k, e)
cmp(ref, again, "eager vs eager:")
cmp(ref, rep, "eager vs replay:")
cmp(ref, rep2, "eager vs replay2:")
cmp(ref, rep3, "eager vs replay3:")
ts.forward_backward(); e3 = snapshot()
cmp(ref, e3, "eager vs eager-after-replays:")
