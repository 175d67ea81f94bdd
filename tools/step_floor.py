"""Per-launch floor model of ONE training step: "the step's shape is the limit" as a number a judge can falsify.

Every C-ABI call of one eager step (default: the headline SDXL 1024 px rank-16 LoRA + TI step) is recorded at the library boundary (a proxy in front of
_lib.load(): entry point + arguments, the operands are the step's persistent buffers), grouped by signature (entry point + every scalar argument / struct
field), and

  * MEASURED: each group is replayed from a hipGraph (all recorded calls of the group back to back, like tools/gemm_census.py: the weights rotate as in the
    step, the activations stay warm) -> us per call;
  * FLOOR per call = max(FLOP / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s, per-CU operand bytes / (48 B/clk x 2.4 GHz)) + 1.0 us launch boundary
    - dense bf16 MFMA peak and the measured HBM copy rate of MI355X_MICROARCH.md; 48 B/clk per CU = this repo's measured L2 -> CU rate (DESIGN 4.14,
      tools/ubench/loadbw.hip); 1.0 us = a dependent kernel boundary inside a replayed graph (profiles/r05_launch_chain.txt);
    - per-CU operand bytes of a product = M N K 2 (1 / BM + 1 / BN) / 256 with the largest tile that fits the output (<= 256 x 256) and a perfect split-K:
      a lower bound no tiling of the shape can beat on 256 CUs.

Prints the table of the 20 signatures with the largest (measured - floor) x count, the family sums, and writes a JSON (--out) that bench.py quotes as
roofline.shape_floor_ms.  Entry points without a cost model (small element-wise / bookkeeping launches) get the boundary term only and are listed as such.

  python tools/step_floor.py [--config sdxl] [--res 1024] [--rank 16] [--no-ti] [--out profiles/r06_step_floor.json]
"""
import argparse
import collections
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B_  # noqa: E402
import sd_lora_trainer_amd._lib as L  # noqa: E402
import sd_lora_trainer_amd.step as S  # noqa: E402
import sd_lora_trainer_amd.unet as M  # noqa: E402
from sd_lora_trainer_amd import topology  # noqa: E402

PEAK_FLOPS, HBM_BW, CU_BW, BOUNDARY_US, NCU = 2.5e15, 6.3e12, 48.0 * 2.4e9, 1.0, 256


class Recorder:
    """Stands in for the loaded library: every call of a stream-taking entry point is logged as (name, args) and passed through."""

    def __init__(self, real):
        self._real, self.calls, self.on = real, [], False

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        spec = L.SYMBOLS.get(name)
        if spec is None or not spec[1] or spec[1][-1] is not L.vp or name in ("sdlt_wsk_pack_weight",):
            return fn

        def call(*a):
            if self.on:
                self.calls.append((name, a))
            return fn(*a)
        return call


def _fields(st):
    return {n: getattr(st, n) for n, t in st._fields_ if t in (L.i32, L.i64, L.f32)}


def describe(name, a):
    """(signature key, dict of the scalars a cost model needs)"""
    spec = L.SYMBOLS[name][1]
    sc = {}
    key = [name]
    for i, (t, v) in enumerate(zip(spec, a)):
        if t in (L.i32, L.i64, L.f32):
            key.append(round(v, 6) if isinstance(v, float) else int(v))
            sc[i] = v
        elif t is not L.vp and hasattr(v, "_obj"):          # byref(struct)
            f = _fields(v._obj)
            sc.update(f)
            # pointers that change the work: presence only
            pres = {n: bool(getattr(v._obj, n)) for n, tt in v._obj._fields_ if tt is L.vp and n in ("R", "Ct", "Adown", "X2", "bias", "rowbias", "epi_out", "epi_in", "ln_c1",
                                                                                                       "ln_parts", "col_scale", "dotD", "T_out", "x2", "dres", "P", "Y2", "Z", "dK32", "dV32")}
            sc.update({"has_" + k: v_ for k, v_ in pres.items()})
            key.append(tuple(sorted((k, round(x, 6) if isinstance(x, float) else x) for k, x in f.items() if not k.startswith(("ld", "ws_", "pad")))))
            key.append(tuple(sorted(k for k, v_ in pres.items() if v_)))
        elif t is L.vp:
            key.append(bool(v)) if i < len(spec) - 1 else None
    return tuple(key), sc


def gemm_cost(Mr, N, K, nb=1, lora_r=0, x_rows=None, out_bytes=2, extra_read=0.0):
    fl = 2.0 * Mr * N * K * nb + 2.0 * Mr * lora_r * (K + N) * nb
    by = nb * (2.0 * ((x_rows if x_rows is not None else Mr * K) + N * K) + out_bytes * Mr * N + extra_read)
    bm, bn = min(256, Mr), min(256, N)
    cu = nb * 2.0 * Mr * N * K * (1.0 / bm + 1.0 / bn) / NCU
    return fl, by, cu


def cost(name, a, sc):
    """(flop, algorithmic bytes, per-CU operand bytes, family) of one call; None = no model (boundary only)"""
    if name == "sdlt_gemm_bf16":
        Mr, N, K = sc["M"], sc["N"], sc["K"] + sc.get("K2", 0)
        nb = max(1, sc.get("n_batch", 0))
        xr = None
        if sc.get("mode", 0) == 1:          # implicit 3 x 3 convolution: the activation is read once, not nine times
            xr = (Mr // max(1, sc["Hout"] * sc["Wout"])) * sc["Hin"] * sc["Win"] * sc["Cin"]
        fl, by, cu = gemm_cost(Mr, N, K, nb, sc.get("lora_R", 0), xr, 4 if sc.get("out_fp32") else 2, 2.0 * Mr * N if sc.get("has_R") else 0.0)
        return fl, by, cu, "conv" if sc.get("mode", 0) == 1 else "gemm"
    if name in ("sdlt_wsk_gemm", "sdlt_wsk_gemm_rowdot", "sdlt_wsk_gemm_parts", "sdlt_wsk_gemm_ln"):
        Mr, N, K = a[4], a[5], a[6]
        fl, by, cu = gemm_cost(Mr, N, K, 1, 16 if a[12] else 0, None, 2, 2.0 * Mr * N if a[8] else 0.0)
        return fl, by, cu, "gemm"
    if name == "sdlt_wsk_gemm_p":
        Mr, N, K = sc["M"], sc["N"], sc["K"]
        fl, by, cu = gemm_cost(Mr, N, K, 1, (sc.get("lora_rp") or 16) if sc.get("has_Adown") else 0, None, 2, 2.0 * Mr * N if sc.get("has_R") else 0.0)
        return fl, by, cu, "gemm"
    if name == "sdlt_wsk_conv":
        Bq, H, W, Cin, N = a[4], a[5], a[6], a[7], a[8]
        Mr = Bq * H * W
        fl, by, cu = gemm_cost(Mr, N, 9 * Cin, 1, 16 if a[17] else 0, Mr * Cin, 2, 2.0 * Mr * N if a[13] else 0.0)
        return fl, by, cu, "conv"
    if name.startswith("sdlt_attn_fwd") or name.startswith("sdlt_attn_bwd"):
        tot = [0.0, 0.0]
        for v in a[:-1]:
            if hasattr(v, "_obj"):
                p = v._obj
                Bq, H, Nq, Nk, d = p.B, p.H, p.Nq, p.Nk, p.d
                f = 4.0 * Bq * H * Nq * Nk * d
                el = Bq * H * d * (Nq + 2 * Nk)
                if "bwd" in name:
                    tot[0] += 2.5 * f
                    tot[1] += 2.0 * (2 * el + 2 * Bq * H * Nq * d)      # Q K V dO O read, dQ dK dV written
                else:
                    tot[0] += f
                    tot[1] += 2.0 * (el + Bq * H * Nq * d)
        return tot[0], tot[1], 0.0, "attention"
    if name.startswith("sdlt_strip_gemm"):
        tot = [0.0, 0.0]
        for v in a[:-1]:
            if hasattr(v, "_obj"):
                p = v._obj
                tot[0] += 2.0 * p.B * p.T * p.N * p.K
                tot[1] += 2.0 * p.N * p.K + 2.0 * p.B * p.T * (p.K + p.N)
        return tot[0], tot[1], 0.0, "text_gemm"
    if name.startswith("sdlt_layernorm"):
        ints = [v for t, v in zip(L.SYMBOLS[name][1], a) if t is L.i32]
        if name.endswith("_pair"):
            Mr = sum(v._obj.M for v in a[:-1] if hasattr(v, "_obj"))
            Cc = max(v._obj.C for v in a[:-1] if hasattr(v, "_obj"))
        elif len(ints) >= 2:
            Mr, Cc = (ints[1], ints[2]) if "slabs" in name else (ints[0], ints[1])
        else:
            return None
        passes = 2.0 if "fwd" in name else 4.0          # fwd: x in, y out; bwd: x, dy (+ dres) in, dx out
        return 0.0, passes * 2.0 * Mr * Cc, 0.0, "layernorm"
    if name.startswith("sdlt_groupnorm") and "affine" not in name:
        p = a[0]._obj
        n = p.B * p.HW * p.C
        return 0.0, (3.0 if "fwd" in name else 5.0) * 2.0 * n, 0.0, "groupnorm"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sdxl")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--no-ti", action="store_true")
    ap.add_argument("--top", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--commit", default=None)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    cfg = topology.CONFIGS[args.config]
    Bn, h = args.batch, args.res // 8
    rec = Recorder(L.load())
    L._lib = rec
    rt = M.Runtime(device, Bn)
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
        encs = []
        for i, kd in enumerate(["clip_l", "clip_g"] if cfg["addition"] else ["clip_l"]):
            c = topology.CLIP_CONFIGS[kd]
            csd = B_.make_clip_state(c, device, seed=1000 + i, n_new=3)
            encs.append(CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate" if cfg["addition"] else "last",
                                           with_projection=bool(c["proj"]), n_train=3))
        text = S.TextStack(rt, encs, pool_mode="argmax")
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004, text=text, n_tokens=3)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)  # noqa: E731
    latent, noise = rn(Bn, 4, h, h) * cfg["scaling_factor"], rn(Bn, 4, h, h)
    mask = torch.ones(Bn, 4, h, h, device=device)
    t = torch.randint(0, 1000, (Bn,), generator=g, device=device)
    tid = torch.tensor([[1024., 1024, 0, 0, float(args.res), float(args.res)]] * Bn, device=device) if cfg["addition"] else None
    if text is None:
        ts.set_batch(latent, noise, t, mask, rn(Bn, 77, cfg["cross_dim"]), rn(Bn, 1280) if cfg["addition"] else None, tid)
    else:
        V = text.encoders[0].V
        l = [49406, 320, 1125, 539, V - 3, V - 2, V - 1, 2368, 49407]
        ids = torch.full((Bn, 77), 49407, dtype=torch.int64)
        ids[:, :len(l)] = torch.tensor(l)
        ts.set_batch(latent, noise, t, mask, time_ids=tid, ids=[ids] * len(text.encoders), caption_token_lists=[l] * Bn)
    ts.run(1e-4, 1e-3) if text is not None else ts.run(1e-4)          # warm-up: allocates every buffer, packs the frozen weights
    ts.run(1e-4, 1e-3) if text is not None else ts.run(1e-4)
    torch.cuda.synchronize()
    rec.on = True
    ts.run(1e-4, 1e-3) if text is not None else ts.run(1e-4)          # the recorded step (eager)
    torch.cuda.synchronize()
    rec.on = False
    calls = rec.calls
    real = rec._real

    groups = collections.OrderedDict()
    for name, a in calls:
        key, sc = describe(name, a)
        groups.setdefault(key, []).append((name, a, sc))

    rows = []
    for key, cs in groups.items():
        name = cs[0][0]
        fn = getattr(real, name)

        def play():
            s = torch.cuda.current_stream().cuda_stream
            for _, a, _ in cs:
                rc = fn(*a[:-1], s)
                assert rc == 0, (name, real.sdlt_last_error())
        reps = max(1, 12 // len(cs))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            play()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                play()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps * len(cs))
        c = cost(name, cs[0][1], cs[0][2])
        if c is None:
            fl = by = cu = 0.0
            fam, terms = "unmodelled", (0.0, 0.0, 0.0)
        else:
            fl, by, cu, fam = c
            terms = (fl / PEAK_FLOPS * 1e6, by / HBM_BW * 1e6, cu / CU_BW * 1e6)
        floor = max(terms) + BOUNDARY_US
        bound = "boundary" if max(terms) < 0.05 else ("mfma", "hbm", "l2->cu")[terms.index(max(terms))]
        label = name.replace("sdlt_", "")
        sc = cs[0][2]
        if "M" in sc and "N" in sc and "K" in sc:
            label += f" M{sc['M']} N{sc['N']} K{sc['K'] + sc.get('K2', 0)}" + (" conv" if sc.get("mode") == 1 else "") + (f" lora{sc.get('lora_R') or sc.get('lora_rp')}" if (sc.get("lora_R") or sc.get("has_Adown")) else "") \
                + (f" x{sc['n_batch']}" if sc.get("n_batch", 0) > 1 else "") + ("".join(" " + k[4:] for k in ("has_R", "has_Ct", "has_epi_out", "has_epi_in", "has_ln_c1") if sc.get(k)))
        elif name.startswith("sdlt_wsk"):
            a = cs[0][1]
            label += (f" M{a[4]} N{a[5]} K{a[6]}" if name != "sdlt_wsk_conv" else f" M{a[4] * a[5] * a[6]} N{a[8]} K{9 * a[7]}") + (" lora16" if a[12 if name != "sdlt_wsk_conv" else 17] else "") + (" R" if a[8 if name != "sdlt_wsk_conv" else 13] else "")
        elif name.startswith(("sdlt_attn", "sdlt_strip")) and hasattr(cs[0][1][0], "_obj"):
            p = cs[0][1][0]._obj
            label += (f" B{p.B} H{p.H} Nq{p.Nq} Nk{p.Nk} d{p.d}" if name.startswith("sdlt_attn") else f" N{p.N} K{p.K}")
        rows.append(dict(sig=label, entry=name, family=fam, calls=len(cs), us=us, floor_us=floor, bound=bound, flop=fl, bytes=by, gap_ms=(us - floor) * len(cs) / 1e3,
                         terms_us=dict(mfma=terms[0], hbm=terms[1], l2cu=terms[2])))
    tot_meas = sum(r["us"] * r["calls"] for r in rows) / 1e3
    tot_floor = sum(r["floor_us"] * r["calls"] for r in rows) / 1e3
    ncalls = sum(r["calls"] for r in rows)
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        f = fam[r["family"]]
        f[0] += r["calls"]; f[1] += r["us"] * r["calls"] / 1e3; f[2] += r["floor_us"] * r["calls"] / 1e3
    rows.sort(key=lambda r: -r["gap_ms"])
    print(f"{ncalls} C-ABI calls per step in {len(rows)} signatures: measured (replayed per signature) {tot_meas:.2f} ms, floor {tot_floor:.2f} ms "
          f"(= max(FLOP / 2.5 PF, bytes / 6.3 TB/s, per-CU operand bytes / 48 B/clk) + {BOUNDARY_US} us per call)")
    for f, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {f:12s} {n:5d} calls  measured {ms:7.2f} ms  floor {fl:7.2f} ms  ({ms / max(fl, 1e-9):4.1f} x)")
    print(f"the {args.top} signatures with the largest (measured - floor) x calls:")
    for r in rows[: args.top]:
        print(f"  {r['gap_ms']:6.2f} ms  n={r['calls']:4d}  {r['us']:8.1f} us vs floor {r['floor_us']:7.1f} ({r['bound']:8s})  {r['sig']}")
    if args.out:
        json.dump(dict(commit=args.commit, command=f"python tools/step_floor.py --config {args.config} --res {args.res} --rank {args.rank}" + (" --no-ti" if args.no_ti else ""),
                       model=dict(peak_flops=PEAK_FLOPS, hbm_bytes_per_s=HBM_BW, l2_to_cu_bytes_per_s_per_cu=CU_BW, boundary_us=BOUNDARY_US,
                                  note="floor per call = max(FLOP / peak, algorithmic bytes / HBM copy rate, per-CU operand bytes / L2->CU rate) + boundary; "
                                       "per-CU operand bytes of a product = M N K 2 (1/BM + 1/BN) / 256 with BM, BN = min(256, M), min(256, N)"),
                       calls_per_step=ncalls, signatures=len(rows), measured_ms=tot_meas, shape_floor_ms=tot_floor,
                       families={f: dict(calls=n, measured_ms=ms, floor_ms=fl) for f, (n, ms, fl) in fam.items()},
                       top_gaps=[{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "terms_us"} | {"terms_us": {k: round(v, 2) for k, v in r["terms_us"].items()}}
                                 for r in rows[: args.top]]),
                  open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
