"""Per-shape census of the GEMM launches of ONE training step (default: the headline SDXL 1024 px LoRA + TI step): every
`ops.gemm` call of an eager step is recorded with its real operands (persistent buffers), grouped by signature, and each group
is replayed from a hipGraph (all recorded calls of the group back to back -> the weights rotate like in the step, activations
stay warm) to get its per-launch time.  Prints count x time, TFLOP/s and the share of the step per signature.

  python tools/gemm_census.py [--config sdxl] [--res 1024] [--no-ti] [--full-ft]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B_  # noqa: E402
import sd_lora_trainer_amd.step as S  # noqa: E402
import sd_lora_trainer_amd.unet as M  # noqa: E402
from sd_lora_trainer_amd import ops, topology  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sdxl")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--no-ti", action="store_true")
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--pmc-order", default=None, help="PMC mode (tools/gemm_fetch_ratio.sh): instead of timing, launch every signature's calls eagerly once more, "
                    "a marker kernel (torch.cuda._sleep) in front of each signature, and write the signature order + algorithmic operand bytes to this JSON file")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    cfg = topology.CONFIGS[args.config]
    B, h = args.batch, args.res // 8
    rt = M.Runtime(device, B)
    g = torch.Generator(device=device).manual_seed(100)
    sd = B_.make_state(cfg, device, seed=0)
    unet = M.UNet(rt, cfg, sd, lora_rank=args.rank)
    for e in unet.arena.entries:
        e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=device) / args.rank)
        e["B"].normal_(0, 0.01, generator=g)
    unet.arena.refresh_shadows()
    del sd
    text = None
    if not args.no_ti:
        import sd_lora_trainer_amd.clip as CL
        kinds = ["clip_l", "clip_g"] if cfg["addition"] else ["clip_l"]
        encs = []
        for i, kd in enumerate(kinds):
            c = topology.CLIP_CONFIGS[kd]
            csd = B_.make_clip_state(c, device, seed=1000 + i, n_new=3)
            encs.append(CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate" if cfg["addition"] else "last",
                                           with_projection=bool(c["proj"]), n_train=3))
        text = S.TextStack(rt, encs, pool_mode="argmax")
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), text=text, n_tokens=3)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)  # noqa: E731
    latent, noise = rn(B, 4, h, h) * cfg["scaling_factor"], rn(B, 4, h, h)
    mask = torch.ones(B, 4, h, h, device=device)
    t = torch.randint(0, 1000, (B,), generator=g, device=device)
    tid = torch.tensor([[1024., 1024, 0, 0, float(args.res), float(args.res)]] * B, device=device) if cfg["addition"] else None
    if text is None:
        ts.set_batch(latent, noise, t, mask, rn(B, 77, cfg["cross_dim"]), rn(B, 1280) if cfg["addition"] else None, tid)
    else:
        V = text.encoders[0].V
        l = [49406, 320, 1125, 539, V - 3, V - 2, V - 1, 2368, 49407]
        ids = torch.full((B, 77), 49407, dtype=torch.int64)
        ids[:, :len(l)] = torch.tensor(l)
        ts.set_batch(latent, noise, t, mask, time_ids=tid, ids=[ids] * len(text.encoders), caption_token_lists=[l] * B)
    ts.body()                       # warm-up: allocates every buffer
    torch.cuda.synchronize()

    calls = []
    real = ops.gemm

    def hook(X, W, out, **kw):
        calls.append((X, W, out, kw))
        return real(X, W, out, **kw)
    ops.gemm = hook
    M._ops.gemm = hook
    ts.body()
    torch.cuda.synchronize()
    ops.gemm = real
    M._ops.gemm = real

    def sig(X, W, out, kw):
        N, K = W.shape
        conv = kw.get("conv")
        Mr = X.shape[0] if conv is None else conv.B * conv.Hout * conv.Wout
        lo = kw.get("lora")
        nb = kw["batch"].n if kw.get("batch") is not None else 1
        cv = "" if conv is None else f"conv(s{conv.stride}u{conv.ups}f{conv.flip}t{conv.tr})"
        extra = "".join([" +K2" if kw.get("X2") is not None else "", f" lora{lo[1].shape[1] if kw.get('lora_group_k', 0) == 0 else 16}" if lo is not None else "",
                         f" gN{kw['lora_group_n']}" if kw.get("lora_group_n") else "", f" gK{kw['lora_group_k']}" if kw.get("lora_group_k") else "",
                         " res" if kw.get("residual") is not None else "", " f32" if (out is not None and out.dtype == torch.float32) else "", " geglu-bwd" if kw.get("geglu_bwd") is not None else "", " geglu" if kw.get("geglu_out") is not None else "", " Ct" if kw.get("Ct") is not None else "",
                         f" x{nb}" if nb > 1 else ""])
        K2 = kw["X2"].shape[1] if kw.get("X2") is not None else 0
        return (Mr, N, K + K2, cv + extra), 2.0 * Mr * N * (K + K2) * nb

    groups = collections.OrderedDict()
    for c in calls:
        s, fl = sig(*c)
        groups.setdefault(s, [[], fl])[0].append(c)
    if args.pmc_order:
        import json
        order = []
        for s, (cs, fl) in groups.items():
            torch.cuda._sleep(1000)                  # marker dispatch: the counter rows between two markers belong to one signature
            fetch_b = write_b = 0.0
            for (X, W, out, kw) in cs:
                real(X, W, out, **kw)
                nb = kw["batch"].n if kw.get("batch") is not None else 1
                conv = kw.get("conv")
                x_el = X.numel() if conv is not None else X.shape[0] * X.shape[1]       # (a 3x3 conv reads its input once, not the 9 taps)
                fetch_b += 2.0 * nb * (x_el + W.shape[0] * W.shape[1] + (kw["X2"].numel() if kw.get("X2") is not None else 0)
                                       + (s[0] * s[1] if kw.get("residual") is not None else 0))
                write_b += nb * s[0] * s[1] * (4.0 if (out is not None and out.dtype == torch.float32) else 2.0)
            order.append(dict(sig=f"M{s[0]} N{s[1]} K{s[2]} {s[3]}".strip(), calls=len(cs), algorithmic_fetch_bytes=fetch_b / len(cs), algorithmic_write_bytes=write_b / len(cs),
                              flop=fl))
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        json.dump(order, open(args.pmc_order, "w"), indent=1)
        return
    rows = []
    for s, (cs, fl) in groups.items():
        reps = max(1, 12 // len(cs))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for (X, W, out, kw) in cs:
                real(X, W, out, **kw)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                for (X, W, out, kw) in cs:
                    real(X, W, out, **kw)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps * len(cs))
        rows.append((us * len(cs), len(cs), us, fl / us / 1e6, s))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    totfl = sum(r[1] * groups[r[4]][1] for r in rows)
    print(f"{len(calls)} gemm launches, {len(rows)} signatures, sum of per-launch times {tot / 1e3:.2f} ms, {totfl / 1e12:.2f} TFLOP -> {totfl / tot / 1e6:.0f} TFLOP/s")
    acc = 0.0
    for (t_us, n, us, tf, s) in rows[: args.top]:
        acc += t_us
        print(f"{t_us / 1e3:7.3f} ms {100 * t_us / tot:5.1f}% (cum {100 * acc / tot:5.1f}%)  n={n:4d}  {us:8.1f} us  {tf:6.0f} TF/s   M{s[0]} N{s[1]} K{s[2]} {s[3]}")


if __name__ == "__main__":
    main()
