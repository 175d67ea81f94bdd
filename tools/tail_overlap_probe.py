"""Can the adapter-gradient / AdamW tail of the UNet run BESIDE the text encoders' backward?  ("if a fork / join ever costs less than it saves")

After the UNet's backward the headline step has two independent chains:
  A  text-encoder backward (2.0 ms of < 256-workgroup launches) -> token-row AdamW
  B  grouped adapter gradients (0.6 ms) -> UNet AdamW -> operand refresh (0.25 ms)
Inside ONE hipGraph a fork / join costs more than B (ROCm's executor, §7).  This probe cuts the step into three graphs instead and joins them with events:
  stream 1:  G1 = text forward + UNet forward / backward | record e1 | G2 = chain A            | wait e2
  stream 2:                                               wait e1   | G3 = chain B | record e2
and times  (a) the product's one graph,  (b) G1, G3, G2 back to back on one stream,  (c) the overlapped form - same session, alternated; (b) and (c) must leave
bit-identical parameters.

  python tools/tail_overlap_probe.py [--steps 40] [--rounds 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sd_lora_trainer_amd.clip as CL  # noqa: E402
import sd_lora_trainer_amd.step as S  # noqa: E402
import sd_lora_trainer_amd.unet as M  # noqa: E402
from sd_lora_trainer_amd import topology  # noqa: E402


def build(device, version="sdxl", res=1024, rank=16):
    """The headline job exactly as bench.py's build_job(0) makes it."""
    cfg = topology.CONFIGS[version]
    B, h, seed, n_tok = 1, res // 8, 0, 3
    rt = M.Runtime(device, B)
    g = torch.Generator(device=device).manual_seed(100 + seed)
    unet = M.UNet(rt, cfg, bench.make_state(cfg, device, seed=seed), lora_rank=rank)
    for e in unet.arena.entries:
        e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=device) / rank)
        e["B"].zero_()
    unet.arena.refresh_shadows()
    encs = []
    for i, kd in enumerate(["clip_l", "clip_g"]):
        c = topology.CLIP_CONFIGS[kd]
        csd = bench.make_clip_state(c, device, seed=1000 + 10 * seed + i, n_new=n_tok)
        encs.append(CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate", with_projection=bool(c["proj"]), n_train=n_tok))
    text = S.TextStack(rt, encs, pool_mode="argmax")
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004, text=text, n_tokens=n_tok)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)  # noqa: E731
    latent, noise = rn(B, 4, h, h) * cfg["scaling_factor"], rn(B, 4, h, h)
    mask = (torch.rand(B, 1, h, h, generator=g, device=device) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    timesteps = torch.randint(0, 1000, (B,), generator=g, device=device)
    rn(B, 77, cfg["cross_dim"]); rn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])      # (bench.py's draws, kept so that the batch is the same)
    tid = torch.tensor([[1024., 1024, 0, 0, float(res), float(res)]] * B, device=device)
    vocab = text.encoders[0].V
    tok = [vocab - 3, vocab - 2, vocab - 1]
    ids = torch.full((B, 77), 49407, dtype=torch.int64)
    gw = torch.Generator().manual_seed(7000 + seed)
    words = torch.randint(1000, min(40000, vocab - 10), (8,), generator=gw).tolist()
    l = [49406] + words[:4] + tok + words[4:] + [49407]
    ids[0, :len(l)] = torch.tensor(l)
    ts.set_batch(latent, noise, timesteps, mask, time_ids=tid, ids=[ids] * 2, caption_token_lists=[l])
    ts.capture(warmup=2)
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ts = build(dev)
    u = ts.unet
    a = ts.group
    state = [a.params] + (a.opt_state() if hasattr(a, "opt_state") else [a.m, a.v]) + [ts.ti.params, ts.ti.m, ts.ti.v]
    s1 = torch.cuda.current_stream(dev)
    s2 = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)

    def cap(fns, pool=None):
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, stream=side):
            for fn in fns:
                fn()
        torch.cuda.current_stream().wait_stream(side)
        return g

    snap = [t.clone() for t in state]
    step0 = ts.opt_step
    with ts._ws():
        real_lg = u.lora_grads
        u.lora_grads = lambda: None
        try:
            g1 = cap([ts._phase_text_fwd, ts._phase_unet])
        finally:
            u.lora_grads = real_lg
        g3 = cap([u.lora_grads, ts._unet_optimizer])                      # its own pool: it runs beside G2
        g2 = cap([ts._phase_text_bwd, ts._other_optimizers], pool=g1.pool())

    def restore():
        for t, c in zip(state, snap):
            t.copy_(c)
        a.refresh_shadows()
        ts.ti.refresh_tables()
        ts.opt_step = step0
        torch.cuda.synchronize()

    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    total = 1000

    def hyper(i):
        ts.set_hyper(bench.lr_at(i, total), 1e-3 * (1 - i / total) ** 1.7, 0.0)

    def one_graph(i):
        ts.run(bench.lr_at(i, total), 1e-3 * (1 - i / total) ** 1.7)

    def serial(i):
        with ts._ws():
            hyper(i)
            g1.replay(); g3.replay(); g2.replay()

    def overlapped(i):
        with ts._ws():
            hyper(i)
            g1.replay()
            e1.record(s1)
            s2.wait_event(e1)
            with torch.cuda.stream(s2):
                g3.replay()
                e2.record(s2)
            g2.replay()
            s1.wait_event(e2)

    def timed(fn, n):
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(n):
            fn(5 + i)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / n, (time.perf_counter() - t0) / n * 1e3

    # same trajectory from the same state: the three forms must agree bit for bit (same kernels on the same operands, only the order of independent launches differs)
    sums = {}
    for name, fn in (("one graph", one_graph), ("three graphs, one stream", serial), ("three graphs, tail beside the text backward", overlapped)):
        restore()
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        sums[name] = (a.params.double().sum().item(), a.params.double().abs().sum().item(), ts.ti.params.double().sum().item(), float(ts.loss.sum().item()) if hasattr(ts, "loss") else 0.0)
    ref = sums["one graph"]
    for k, v in sums.items():
        print(f"{k:48s} params sum {v[0]:+.12e} |.| {v[1]:.12e} token rows {v[2]:+.12e}  {'== one graph' if v == ref else 'DIFFERS'}")
    for r in range(args.rounds):
        for name, fn in (("one graph", one_graph), ("three graphs, one stream", serial), ("three graphs, tail beside the text backward", overlapped)):
            restore()
            ev, wall = timed(fn, args.steps)
            print(f"round {r}: {name:48s} {ev:7.3f} ms/step (events)  {wall:7.3f} (host clock)", flush=True)


if __name__ == "__main__":
    main()
