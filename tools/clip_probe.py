"""Text-encoder stack alone (CLIP-L + bigG, random weights): graph-replayed forward+backward, encoders serial vs on side streams."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sd_lora_trainer_amd import topology, unet as M, step as S, clip as CL
dev = "cuda:0"
rt = M.Runtime(dev, 1)
encs = []
for i, kd in enumerate(["clip_l", "clip_g"]):
    c = topology.CLIP_CONFIGS[kd]
    csd = bench.make_clip_state(c, dev, seed=1000 + i, n_new=3)
    encs.append(CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate", with_projection=bool(c["proj"]), n_train=3))
text = S.TextStack(rt, encs, pool_mode="argmax")
ids = torch.randint(1000, 40000, (1, 77)); ids[0, 0] = 49406; ids[0, 20:] = 49407
text.set_ids([ids.to(dev)] * 2)
ctx = rt.zeros(M.CTX_PAD, 2048); dctx = (torch.randn(M.CTX_PAD, 2048, device=dev) * 0.01).to(rt.act); dctx[77:] = 0
dpool = (torch.randn(1, 1280, device=dev) * 0.01).to(rt.act)
grads = [torch.zeros(3, e.D, device=dev) for e in encs]
def body(which):
    if which in ("fwd", "both"): text.forward(ctx)
    if which in ("bwd", "both"): text.backward(dctx, dpool, grads)
sides = text.side
for par in (False, True):
    text.side = sides if par else []
    for which in ("fwd", "bwd", "both"):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body("both")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(which)
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(f"parallel={par} {which:5s} {(time.perf_counter() - t) / 10 * 1e3:.3f} ms")
if len(sys.argv) > 1:   # per-encoder
    text.side = []
    for e, w in zip(encs, (768, 1280)):
        gph = torch.cuda.CUDAGraph()
        off = 0 if w == 768 else 768
        f = lambda: (e.forward(text.ids[0], 1, hidden_out=ctx[:, off:off + w], pool_rows=text.pool_rows), e.backward(dctx[:, off:off + w], dpool if e.with_projection else None, grads[0 if w == 768 else 1]))
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(gph): f()
        gph.replay(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): gph.replay()
        torch.cuda.synchronize(); print(f"encoder D={w}: fwd+bwd {(time.perf_counter() - t) / 10 * 1e3:.3f} ms")
