"""Headline benchmark: training images/sec, SDXL 1024x1024 (latent 128x128), rank-16 LoRA, batch 1 per GPU,
job-parallel over N GPUs (BASELINE.json metric; reference loop /root/reference main.py:263-382).

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU; ranks are INDEPENDENT training jobs - the
   hyper-parameter-sweep workload of scripts/create_hyperparam_sweep.py - so there is no data-path collective;
   RCCL is only used for the timing barrier and the max-over-ranks reduction.)

A "step" = one full optimisation step on one synthetic batch, inputs resident in HBM: DDPM add-noise, UNet+LoRA
forward, masked/min-SNR MSE, backward to every LoRA A/B and to the text conditioning, fused AdamW(+L1), bf16 shadow
refresh - one hipGraph replay.  Synthetic data / random-init weights of the exact architecture (no network here).

Prints ONE JSON line.  roofline: the step is a dense contraction -> MFMA bound; achieved = algorithmic FLOPs of one
step (2 x forward census, SURVEY.md 8d: 13.66 TFLOP for this config) / measured step time (HIP events on the stream
the graph is replayed on); peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).  cpu_baseline: the fp32 oracle
(oracle/, a port of the reference path) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_DENSE = 2.5e15


def make_state(cfg, device, seed):
    """Random-init weights of the exact architecture, generated directly in HBM."""
    from sd_lora_trainer_amd import topology
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for n, shp in topology.param_shapes(cfg).items():
        t = torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        if len(shp) >= 2:
            t *= 1.0 / math.sqrt(math.prod(shp[1:]))
        else:
            t *= 0.02
            if n.endswith(".weight"):   # norm gains
                t += 1.0
        sd[n] = t
    return sd


def make_clip_state(c, device, seed, n_new):
    from sd_lora_trainer_amd import topology
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for n, shp in topology.clip_param_shapes(c, n_new).items():
        t = torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        if "embedding" in n:
            t *= 0.02
        elif len(shp) == 2:
            t *= 1.0 / math.sqrt(shp[1])
        else:
            t *= 0.02
            if "layer_norm" in n and n.endswith(".weight"):
                t += 1.0
        sd[n] = t
    return sd


def lr_at(step, max_steps, unet_lr=1e-3, base=5e-5):
    """LoRA learning-rate schedule of the reference (main.py:236-240, 268-291; warm-up = max_train_steps)."""
    return base * (unet_lr / base) ** (step / max_steps)


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container that shows 256 CPUs but
    is allowed 32 makes a 256-thread OpenMP team crawl), then halved when SMT siblings are counted twice."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
        if phys:
            n = min(n, phys)
    except Exception:
        pass
    return max(1, n)


class GpuTelemetry:
    """Shader clock / socket power / temperature of ONE GPU sampled from a host thread around a timed region (the
    chip clocks to its power budget, MI355X_MICROARCH.md "DVFS give-back", so two boxes - or a cold and a warm box - time the same kernels
    differently; the bench line says which clock its figure was taken at).  amdsmi when the driver is reachable, the amdgpu sysfs files
    otherwise; every failure degrades to `None` fields - the measurement never depends on it."""

    def __init__(self, index=0, period_s=0.02):
        import threading
        self.period, self.samples, self._stop, self._thr = period_s, [], threading.Event(), None
        self.static, self._read = {}, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            h = hs[min(index, len(hs) - 1)]
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(h)
                self.static["power_cap_W"] = float(cap.get("power_cap", 0)) / (1e6 if float(cap.get("power_cap", 0)) > 1e5 else 1.0)
            except Exception:
                pass
            try:
                ci = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                self.static["gfx_clk_max_MHz"] = ci.get("max_clk")
            except Exception:
                pass

            def read():
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
                clk = (sum(clks) / len(clks)) if clks else m.get("current_gfxclk")
                pw = m.get("current_socket_power")
                if not isinstance(pw, (int, float)) or pw <= 0 or pw > 5000:
                    pw = m.get("average_socket_power")
                return clk, pw, m.get("temperature_hotspot")
            read()
            self._read, self.static["source"] = read, "amdsmi gpu_metrics"
        except Exception:
            self._read = self._sysfs_reader(index)

    def _sysfs_reader(self, index):
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if not cards:
            return None
        dev = os.path.dirname(cards[min(index, len(cards) - 1)])
        hw = (glob.glob(os.path.join(dev, "hwmon", "hwmon*")) or [None])[0]

        def rd(p, scale):
            try:
                return float(open(p).read().split()[0]) / scale
            except Exception:
                return None

        def read():
            clk = None
            try:
                for ln in open(os.path.join(dev, "pp_dpm_sclk")):
                    if ln.strip().endswith("*"):
                        clk = float(ln.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            except Exception:
                pass
            if hw:
                clk = rd(os.path.join(hw, "freq1_input"), 1e6) or clk
                return clk, rd(os.path.join(hw, "power1_average"), 1e6) or rd(os.path.join(hw, "power1_input"), 1e6), rd(os.path.join(hw, "temp2_input"), 1e3)
            return clk, None, None
        try:
            read()
        except Exception:
            return None
        self.static["source"] = "amdgpu sysfs"
        if hw:
            cap = rd(os.path.join(hw, "power1_cap"), 1e6)
            if cap:
                self.static["power_cap_W"] = cap
        return read

    def start(self):
        import threading
        if self._read is None:
            return self
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append(self._read())
                except Exception:
                    pass
                self._stop.wait(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=1.0)
            self._thr = None
        out = dict(self.static)
        out["samples"] = len(self.samples)
        for k, name in enumerate(("gfx_clk_MHz", "socket_power_W", "hotspot_C")):
            v = [s[k] for s in self.samples if isinstance(s[k], (int, float))]
            out[name] = {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)} if v else None
        return out


def run_guarded(argv, timeout_s):
    """Run `python bench.py <argv>` as a child with a wall-clock limit; returns the JSON objects it printed (one per line) - whatever
    was complete when it ended or was killed.  The extras of the default run (CPU baseline, train() loop) must never be able to keep
    the one JSON line from being printed."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=timeout_s).stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    res = []
    for line in out.splitlines():
        if line.startswith("{"):
            try:
                res.append(json.loads(line))
            except Exception:
                pass
    return res


def cpu_baseline(version, rank, full_hw, budget_s=150.0, with_text=True):
    """The fp32 oracle (oracle/: CPU port of the reference path) timed on this host's cores on THE SAME STEP the GPU line times:
    `oracle.step_ref.RefTrainer.step` = main.py:263-382 - text encoders forward + backward down to the token tables (the installed Hugging
    Face CLIP classes at the CLIP-L / OpenCLIP-bigG sizes, random init), add_noise, UNet forward + backward to every LoRA tensor, masked /
    min-SNR MSE, token-attention loss, L1 penalty, std regulariser, gradient masking of the frozen rows, torch.optim.AdamW over the
    LoRA tensors and over the token tables.  (`with_text=False`, the --no-ti workload: injected conditioning, no text encoders.)
    1 warm-up + timed steps at the LARGEST resolution of {full, full/2, full/4} whose three steps fit the time budget (calibrated on a
    quick pass at the smallest one), torch.set_num_threads(all cores).  Returns a dict for the JSON line."""
    from oracle import step_ref as R
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import topology
    cfg = U.CONFIGS[version]
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    sd = {}
    for n, shp in U.param_shapes(cfg).items():          # same statistics as oracle.unet_ref.init_unet_state, drawn multi-threaded
        t = torch.randn(shp)
        t = t / math.sqrt(math.prod(shp[1:])) if len(shp) >= 2 else t * 0.02
        if (".norm" in n or n.startswith("conv_norm_out")) and len(shp) == 1 and n.endswith(".weight"):
            t = 1.0 + t
        sd[n] = t
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.01)
    n_tok, text_models, vocab = 3, None, None
    if with_text:
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
        tiny = version.startswith("tiny")
        kinds = (["tiny_l", "tiny_g"] if tiny else ["clip_l", "clip_g"]) if cfg["addition"] else (["tiny_l"] if tiny else ["clip_l"])
        text_models = []
        for kd in kinds:
            c = topology.CLIP_CONFIGS[kd]
            vocab = c["vocab"] + n_tok
            hc = CLIPTextConfig(vocab_size=vocab, hidden_size=c["width"], intermediate_size=c["mlp"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                                max_position_embeddings=77, hidden_act=c["act"], projection_dim=c["proj"] or 768, eos_token_id=2, bos_token_id=0, pad_token_id=1)
            text_models.append((CLIPTextModelWithProjection if c["proj"] else CLIPTextModel)(hc).eval())
    tr = R.RefTrainer(cfg, sd, lora, text_models=text_models, n_tokens=n_tok, train_ids=[vocab - 3, vocab - 2, vocab - 1] if with_text else None,
                      snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004)

    def one_step(h):
        g = torch.Generator().manual_seed(1)
        latent = torch.randn(1, 4, h, h, generator=g) * cfg["scaling_factor"]
        noise = torch.randn(1, 4, h, h, generator=g)
        mask = torch.rand(1, 1, h, h, generator=g).repeat(1, 4, 1, 1) * 0.95 + 0.05
        t = torch.tensor([500])
        tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]]) if cfg["addition"] else None
        kw = {}
        if with_text:                # the caption of the GPU line: BOS, 8 words, the 3 trained tokens in the middle, EOS, padding
            words = torch.randint(1000 if vocab > 2000 else 10, min(40000, vocab - 10), (8,), generator=g).tolist()
            bos, eos = (49406, 49407) if vocab > 49407 else (vocab - 5, vocab - 4)
            l = [bos] + words[:4] + [vocab - 3, vocab - 2, vocab - 1] + words[4:] + [eos]
            ids = torch.full((1, 77), eos, dtype=torch.int64)
            ids[0, :len(l)] = torch.tensor(l)
            kw = dict(ids=ids, caption_token_lists=[l], time_ids=tid, lr_ti=1e-3)
        else:
            kw = dict(ctx=torch.randn(1, 77, cfg["cross_dim"], generator=g),
                      added_cond={"text_embeds": torch.randn(1, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g), "time_ids": tid} if cfg["addition"] else None)
        t0 = time.time()
        tr.step(latent, noise, t, mask, lr=1e-4, **kw)
        return time.time() - t0

    flops = lambda h: 2 * topology.fwd_flops(topology.CONFIGS[version], 1, h, h, rank)["total"]  # noqa: E731
    what = "text encoders + UNet fwd+bwd + losses + both AdamWs" if with_text else "UNet fwd+bwd + losses + AdamW"
    small = max(full_hw // 4, 8)
    one_step(small)                                      # cold pass (page-in, thread pools, AdamW state)
    t_small = one_step(small)
    print(json.dumps(dict(times=[t_small], hw=small, flops=flops(small), cores=cores, what=what)), flush=True)     # (child mode: a first, complete answer)
    # climb small -> full/2 -> full while the next size's warm-up + 2 timed steps still fit what is left of the budget, each estimate
    # scaled by FLOPs from the LAST size measured (small sizes run at a lower rate, so they over-estimate the big ones)
    h, t_h, times, spent = small, t_small, [t_small], 2.0 * t_small
    for cand in (full_hw // 2, full_hw):
        if cand <= h:
            continue
        if spent + 3.0 * t_h * flops(cand) / flops(h) > budget_s:
            break
        t0 = time.time()
        one_step(cand)                                   # warm-up at this size
        times = [one_step(cand)]
        print(json.dumps(dict(times=times, hw=cand, flops=flops(cand), cores=cores, what=what)), flush=True)      # (complete answer, should the guard cut the next step)
        times.append(one_step(cand))
        h, t_h = cand, sum(times) / 2
        spent += time.time() - t0
        print(json.dumps(dict(times=times, hw=h, flops=flops(h), cores=cores, what=what)), flush=True)
    return dict(times=times, hw=h, flops=flops(h), cores=cores, what=what)


def library_gpu_baseline(version, rank, hw, steps=5):
    """The SAME step as cpu_baseline (oracle.step_ref.RefTrainer.step, the torch restatement of main.py:263-382), executed on cuda:0 by PyTorch-ROCm's
    own kernels the way the reference runs it: frozen UNet / text-encoder weights in bf16 (config.py:99 weight_type), trainable LoRA tensors and
    token tables in fp32, torch.autocast(bfloat16), F.scaled_dot_product_attention for every attention without the DAAM hook (diffusers
    AttnProcessor2_0), torch.optim.AdamW.  A reported baseline like cpu_baseline - what the library path reaches on this GPU for the step the HIP
    path replaces (diffusers / peft are thin module wrappers over these calls; they are not installed here).  Returns a dict for the JSON line."""
    from oracle import step_ref as R
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import topology
    dev = torch.device("cuda", 0)
    cfg = U.CONFIGS[version]
    torch.manual_seed(0)
    sd = {}
    for n, shp in U.param_shapes(cfg).items():
        t = torch.randn(shp, device=dev)
        t = t / math.sqrt(math.prod(shp[1:])) if len(shp) >= 2 else t * 0.02
        if (".norm" in n or n.startswith("conv_norm_out")) and len(shp) == 1 and n.endswith(".weight"):
            t = 1.0 + t
        sd[n] = t.to(torch.bfloat16)
    lora = {k: tuple(t.to(dev) for t in v) for k, v in U.init_lora(cfg, rank, seed=1, b_std=0.01).items()}
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    n_tok, text_models, vocab = 3, [], None
    for kd in (["clip_l", "clip_g"] if cfg["addition"] else ["clip_l"]):
        c = topology.CLIP_CONFIGS[kd]
        vocab = c["vocab"] + n_tok
        hc = CLIPTextConfig(vocab_size=vocab, hidden_size=c["width"], intermediate_size=c["mlp"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                            max_position_embeddings=77, hidden_act=c["act"], projection_dim=c["proj"] or 768, eos_token_id=2, bos_token_id=0, pad_token_id=1)
        m = (CLIPTextModelWithProjection if c["proj"] else CLIPTextModel)(hc).eval().to(dev).to(torch.bfloat16)
        m.get_input_embeddings().weight.data = m.get_input_embeddings().weight.data.float()       # the trained tables stay fp32 (main.py:106-113 upcasts what it trains)
        text_models.append(m)
    tr = R.RefTrainer(cfg, sd, lora, text_models=text_models, n_tokens=n_tok, train_ids=[vocab - 3, vocab - 2, vocab - 1], snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004)
    U.USE_SDPA = True
    g = torch.Generator().manual_seed(1)
    latent = (torch.randn(1, 4, hw, hw, generator=g) * cfg["scaling_factor"]).to(dev)
    noise = torch.randn(1, 4, hw, hw, generator=g).to(dev)
    mask = (torch.rand(1, 1, hw, hw, generator=g).repeat(1, 4, 1, 1) * 0.95 + 0.05).to(dev)
    t = torch.tensor([500], device=dev)
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * hw, 8. * hw]], device=dev) if cfg["addition"] else None
    words = torch.randint(1000, 40000, (8,), generator=g).tolist()
    l = [49406] + words[:4] + [vocab - 3, vocab - 2, vocab - 1] + words[4:] + [49407]
    ids = torch.full((1, 77), 49407, dtype=torch.int64)
    ids[0, :len(l)] = torch.tensor(l)
    ids = ids.to(dev)

    def one():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return tr.step(latent, noise, t, mask, lr=1e-4, ids=ids, caption_token_lists=[l], time_ids=tid, lr_ti=1e-3)
    for _ in range(2):
        out = one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    return {"value": 1.0 / dt, "unit": "images/s", "ms_per_step": dt * 1e3, "steps": steps, "loss": out["tot_loss"],
            "sample": f"the torch restatement of the reference step (oracle.step_ref.RefTrainer.step: text encoders, UNet + LoRA forward / backward with autograd, both losses, L1, "
                      f"both torch.optim.AdamW) on cuda:0 at {hw * 8}x{hw * 8} B=1: bf16 frozen weights, fp32 trained tensors, torch.autocast(bfloat16), scaled_dot_product_attention "
                      f"for the attentions without the DAAM hook - PyTorch-ROCm's library kernels on the same GPU, eager, after 2 warm-up steps"}


def train_loop_measure(args):
    """The workload driven by the train() generator (sd_lora_trainer_amd.train, the main.py:34-551 mirror): per step LR schedules,
    posterior sampling, noise / timestep draws, caption dropout, host->device copies of the batch (set_batch), then the graph replay;
    images_per_second as the loop itself measures it (checkpoint writes and the one-off graph capture excluded)."""
    import shutil
    import tempfile
    from sd_lora_trainer_amd.config import TrainingConfig
    from sd_lora_trainer_amd.train import train
    version = args.config
    res = args.res or (1024 if "xl" in version else 512)
    B = args.batch or (1 if "xl" in version else 4)
    tmp = tempfile.mkdtemp(prefix="sdlt_bench_")
    try:
        n_loop = max(5 * args.steps, 100)            # long enough that the loop's own logging (a host sync every n/20 steps) is as rare as in a real run
        cfg_t = TrainingConfig(lora_training_urls="synthetic:8", concept_mode="object", pretrained_model={"path": f"synthetic:{version}"}, seed=0,
                               resolution=res, train_batch_size=B, max_train_steps=n_loop, lora_rank=args.rank, output_dir=tmp, n_sample_imgs=0,
                               unet_lr=1e-3, ti_lr=1e-3)
        gen = train(cfg_t)
        try:
            while True:
                next(gen)
        except StopIteration as e:
            cfg_done, _ = e.value
        ips = cfg_done.training_attributes["images_per_second"]
        return {"value": ips, "unit": "images/s", "ms_per_step": 1e3 * B / ips, "steps": n_loop + 1,
                "note": "extra measurement, not `value`: the train() generator's own loop on the same workload (synthetic 8-image latent cache), "
                        "host work of every step included"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def relaunch_ranks(n, argv, launch_test=False):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script, one per GPU, under torch.distributed.run on this
    node (the job-parallel sweep of /root/reference scripts/create_hyperparam_sweep.py:135-152 is N processes as well) and pass
    their exit code on.  Fails loudly when the node has fewer than N GPUs - never a silent 1-GPU number under an N-GPU flag."""
    import socket
    import subprocess
    if not launch_test:
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s); refusing to report an {n}-GPU figure")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def launch_test_rank(args):
    """`--launch-test` (CPU, gloo): every rank runs the bench's timing protocol - barrier, K 'steps', barrier, MAX over ranks - on a
    sleep instead of the workload and rank 0 prints the JSON skeleton.  Checks the launcher and the multi-rank contract of the line
    (n_gpus, whole-job value, parallelism) where there is no GPU; never a measurement."""
    from sd_lora_trainer_amd import parallel
    rank, world, _ = parallel.init_distributed("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    parallel.barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    parallel.barrier_sync()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "launch test (no workload)", "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak",
                          "config": {"workload": "sleep", "parallelism": f"job-parallel x{world} ({world} rank(s), no collective)"}}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="sdxl", choices=["sdxl", "sd15", "tinyxl", "tiny15"])
    ap.add_argument("--res", type=int, default=0, help="image resolution (default 1024 for sdxl, 512 for sd15)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--dora", action="store_true", help="weight-decomposed adapters (use_dora: trained magnitudes, no L1 penalty / weight decay, config.py:153-157)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-ti", action="store_true", help="inject the text conditioning instead of running the text encoders + TI")
    ap.add_argument("--ti-frozen", action="store_true", help="time the step after freeze_ti_after_completion_f (ti lr = 0): no text-encoder backward")
    ap.add_argument("--jobs-per-gpu", type=int, default=1, help="independent LoRA jobs stepped concurrently on each GPU (own weights, adapters, "
                    "text encoders and hipGraph each, one stream per job); a 'step' then advances every job once")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the extra two-jobs-per-GPU measurement of the default run")
    ap.add_argument("--no-train-loop", action="store_true", help="skip the extra measurement of the train() generator's own step loop")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)      # child modes of the guarded extras
    ap.add_argument("--train-loop-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--library-gpu-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-library-gpu", action="store_true", help="skip the extra measurement of the torch restatement of the step on the GPU (library kernels)")
    ap.add_argument("--full-ft", action="store_true", help="full-UNet fine-tune (BASELINE configs[4], train_configs/full_finetuning_example.json: "
                    "SDXL 512 px, batch 4 per GPU, AdamW over every UNet parameter); data parallel when --gpus > 1: per-bucket reduce-scatter, sharded AdamW, all-gather (SDLT_DDP_ZERO1=0: all-reduce)")
    ap.add_argument("--fp32-moments", action="store_true", help="--full-ft: AdamW with fp32 moments instead of the example config's AdamW8bit (A/B of the optimizer pass)")
    ap.add_argument("--dry-collectives", action="store_true", help="--full-ft on ONE GPU: run the data-parallel exchange step (per-bucket graphs, in-place reduce-scatter / all-gather, "
                    "async works) on a 1-rank RCCL group - the exact call sequence of --gpus N - and report every collective's bytes and the ring wire time it implies for 2 / 4 / 8 GPUs")
    ap.add_argument("--ddp-wire", default=None, choices=["fp32", "bf16"], help="--full-ft --gpus N: dtype of the matrix gradients on the xGMI wire "
                    "(TrainStep(ddp_wire_dtype=); default fp32 = exact)")
    ap.add_argument("--profile-json", default=None, help="step profile of THIS command (tools/step_profile.py over the rocprofv3 kernel trace + FETCH_SIZE / "
                    "WRITE_SIZE passes): source of roofline.traffic and roofline.hbm_kernels; default: the newest profiles/rNN_sdxl1024_ti_step_profile.json for the default workload")
    ap.add_argument("--families-only", action="store_true", help=argparse.SUPPRESS)          # child mode: per-family kernel time of THIS process's replayed steps (torch.profiler)
    ap.add_argument("--no-families", action="store_true", help="skip the extra in-run measurement of the per-family kernel times (roofline.families_this_run)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the extra sustained-rate measurement behind the timed region")
    ap.add_argument("--sustained-steps", type=int, default=100)
    ap.add_argument("--launch-test", action="store_true", help=argparse.SUPPRESS)            # tests/test_parallel_cpu.py: launcher + timing protocol on CPU
    args = ap.parse_args()
    world_env = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and world_env == 0:
        # not under a launcher: spawn the N ranks ourselves (the driver's `torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE)
        sys.exit(relaunch_ranks(args.gpus, sys.argv[1:], launch_test=args.launch_test))
    if world_env and world_env != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world_env}: the line would misreport n_gpus")
    if args.launch_test:
        launch_test_rank(args)
        return
    if args.cpu_baseline_only:          # child of the default run: prints one JSON object per completed measurement
        version = args.config
        res = args.res or (1024 if "xl" in version else 512)
        print(json.dumps(cpu_baseline(version, args.rank, res // 8, with_text=not args.no_ti)), flush=True)
        return
    if args.library_gpu_only:
        version = args.config
        res = args.res or (1024 if "xl" in version else 512)
        print(json.dumps(library_gpu_baseline(version, args.rank, res // 8)), flush=True)
        return
    if args.train_loop_only:
        print(json.dumps(train_loop_measure(args)), flush=True)
        return

    from sd_lora_trainer_amd import parallel
    rank, world, local_rank = parallel.init_distributed("nccl")
    torch.cuda.set_device(local_rank if world > 1 else 0)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    dry = bool(args.dry_collectives) and args.full_ft and world == 1
    if dry and not torch.distributed.is_initialized():          # a 1-rank RCCL communicator: the collectives are real library calls on this GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    from sd_lora_trainer_amd import topology

    version = args.config
    cfg = topology.CONFIGS[version]
    full_ft = args.full_ft
    res = args.res or (512 if full_ft else (1024 if "xl" in version else 512))
    B = args.batch or (4 if full_ft else (1 if "xl" in version else 4))
    h = res // 8
    J = max(1, args.jobs_per_gpu)
    assert not (full_ft and J > 1), "the full fine-tune is one data-parallel job"

    def build_job(jidx):
        """One independent training job (own weights copy, own adapters, own text encoders, own hipGraph)."""
        seed = rank * 16 + jidx
        rt = M.Runtime(device, B)
        g = torch.Generator(device=device).manual_seed(100 + seed)
        if full_ft:
            from sd_lora_trainer_amd import fullft
            sd = make_state(cfg, device, seed=0)       # data-parallel replicas start from the same weights
            trainer = fullft.WeightTrainer(rt)
            unet = M.UNet(rt, cfg, sd, trainer=trainer)
            arena = trainer
        else:
            sd = make_state(cfg, device, seed=seed)    # every rank = its own independent job
            unet = M.UNet(rt, cfg, sd, lora_rank=args.rank, use_dora=args.dora)
            arena = unet.arena
            for e in arena.entries:   # peft "gaussian" init: A ~ N(0, 1/r), B = 0 at step 0 (optimizer.py:89)
                e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=device) / args.rank)
                e["B"].zero_()
            arena.refresh_shadows()
        del sd
        torch.cuda.empty_cache()
        text, n_tok = None, 3
        clip_flops = 0.0
        if not args.no_ti and not full_ft:             # the full fine-tune example disables textual inversion
            import sd_lora_trainer_amd.clip as CL
            tiny = version.startswith("tiny")
            kinds = (["tiny_l", "tiny_g"] if tiny else ["clip_l", "clip_g"]) if cfg["addition"] else (["tiny_l"] if tiny else ["clip_l"])
            encs = []
            for i, kd in enumerate(kinds):
                c = topology.CLIP_CONFIGS[kd]
                csd = make_clip_state(c, device, seed=1000 + 10 * seed + i, n_new=n_tok)
                mode = "penultimate" if cfg["addition"] else "last"
                enc = CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode=mode, with_projection=bool(c["proj"]), n_train=n_tok)
                encs.append(enc)
                clip_flops += topology.clip_fwd_flops(c, B, layers_run=enc.n_run)
                del csd
            text = S.TextStack(rt, encs, pool_mode="argmax")
        ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0 if args.dora else 0.03, weight_decay=0.0 if args.dora else 0.004, text=text, n_tokens=n_tok,
                         process_group=True if (full_ft and (world > 1 or dry)) else None, ddp_wire_dtype=args.ddp_wire if (full_ft and (world > 1 or dry)) else None,
                         ddp_force=dry, optimizer="AdamW8bit" if (full_ft and not args.fp32_moments) else "adamw")      # full_finetuning_example.json: unet_optimizer_type AdamW8bit
        rn = lambda *s: torch.randn(*s, generator=g, device=device)  # noqa: E731
        latent = rn(B, 4, h, h) * cfg["scaling_factor"]
        noise = rn(B, 4, h, h)
        mask = (torch.rand(B, 1, h, h, generator=g, device=device) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
        timesteps = torch.randint(0, 1000, (B,), generator=g, device=device)
        ctx = rn(B, 77, cfg["cross_dim"])
        pooled = tid = None
        if cfg["addition"]:
            pooled = rn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])
            tid = torch.tensor([[1024., 1024, 0, 0, float(res), float(res)]] * B, device=device)
        if text is None:
            ts.set_batch(latent, noise, timesteps, mask, ctx, pooled, tid)
        else:
            vocab = text.encoders[0].V
            tok = [vocab - 3, vocab - 2, vocab - 1]
            lists, ids = [], torch.full((B, 77), 49407 if vocab > 49407 else vocab - 4, dtype=torch.int64)
            gw = torch.Generator().manual_seed(7000 + seed)      # (host generator, seeded like every other draw: `final_loss` is a function of the code, not of the run)
            for b in range(B):   # "a photo of <s0><s1><s2> ..." style caption: BOS, 8 words, the 3 TI tokens, EOS, padding
                words = torch.randint(1000 if vocab > 2000 else 10, min(40000, vocab - 10), (8,), generator=gw).tolist()
                l = [49406 if vocab > 49407 else vocab - 5] + words[:4] + tok + words[4:] + [49407 if vocab > 49407 else vocab - 4]
                ids[b, :len(l)] = torch.tensor(l)
                lists.append(l)
            ts.set_batch(latent, noise, timesteps, mask, time_ids=tid, ids=[ids] * len(text.encoders), caption_token_lists=lists)
        if not args.no_graph:
            ts.capture(warmup=2)
        return ts, arena, text, clip_flops

    # J independent jobs per GPU, each on its own stream with its own graph: at batch 1 the step is bound by per-kernel latency and
    # partial waves of workgroups, so the replays of two jobs overlap on the GPU (two PROCESSES time-slice instead)
    if J > 1:
        from sd_lora_trainer_amd import ops as _ops
        _ops.set_throughput_hint(True)
    cur = torch.cuda.current_stream(device)
    streams = [cur] if J == 1 else [torch.cuda.Stream(device=device) for _ in range(J)]
    jobs = []
    for j in range(J):
        streams[j].wait_stream(cur)
        with torch.cuda.stream(streams[j]):
            jobs.append(build_job(j))
        streams[j].synchronize()
    ts, arena, text, clip_flops = jobs[0]
    total = args.warmup + args.steps
    ti_lr = 1e-3 if (text is not None and not args.ti_frozen) else 0.0

    def step_all(i):
        for (tsj, _, _, _), st in zip(jobs, streams):
            with torch.cuda.stream(st):
                tsj.run(lr_at(i, total), ti_lr * (1 - i / total) ** 1.7)                            # main.py:271-274

    for i in range(args.warmup):
        step_all(i)
    for st in streams:
        cur.wait_stream(st)

    barrier = parallel.barrier_sync
    tele = GpuTelemetry(index=local_rank if world > 1 else 0) if rank == 0 else None
    barrier()
    if tele:
        tele.start()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for st in streams:
        st.wait_stream(cur)
    for i in range(args.steps):
        step_all(args.warmup + i)
    for st in streams:
        cur.wait_stream(st)
    ev1.record()
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device)
    telemetry = tele.stop() if tele else None
    ev_ms = ev0.elapsed_time(ev1)
    losses = [tsj.total_loss() for tsj, _, _, _ in jobs]
    loss = losses[0]
    assert all(math.isfinite(x) for x in losses), "non-finite loss in the timed region"

    if args.families_only:
        # per-family kernel time of the replayed step IN THIS PROCESS (the families of the bench line used to come from a committed profile of another box):
        # three more steps under torch.profiler (roctracer kernel records), folded with the family rules of tools/step_profile.py
        import re
        from torch.profiler import ProfilerActivity, profile
        nprof = 4
        torch.cuda.synchronize()                 # (nothing of the timed region may still be draining when the tracer starts)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(nprof):
                step_all(args.warmup + (i % max(1, args.steps)))
            torch.cuda.synchronize()

        def fam_of(k):
            k = re.sub(r"\(anonymous namespace\)::", "", k).replace("void ", "")
            if k.startswith(("gemm_kernel", "wsk_kernel")): return "gemm"
            if k.startswith("strip_"): return "text_gemm"
            if k.startswith("attn"): return "attention"
            if "lora_grad" in k: return "lora_grad"
            if k.startswith("gn_"): return "groupnorm"
            if k.startswith("ln_"): return "layernorm"
            if k.startswith("geglu"): return "geglu"
            if k.startswith(("adamw_kernel", "shadow_kernel")): return "adamw"
            if "at::" in k or "rocclr" in k or "Memcpy" in k or "Memset" in k: return "torch"
            return "other"
        # whole steps only: the kernel records between the first and the last once-per-step loss kernel (the tracer does not always deliver the first replay's records)
        evs = sorted(((e.time_range.start, e.name, float(e.device_time)) for e in prof.events()
                      if str(getattr(e, "device_type", "")).endswith("CUDA") and getattr(e, "device_time", 0) > 0), key=lambda t: t[0])
        marks = [i for i, (_, n, _) in enumerate(evs) if "mse_reduce_kernel" in n]
        assert len(marks) >= 1 + J, f"torch.profiler delivered {len(marks)} step markers"
        evs = evs[marks[0] + 1: marks[-1] + 1]
        nprof = (len(marks) - 1) // J
        fam_us, fam_n = {}, {}
        for _, name, dur in evs:
            f = fam_of(name)
            fam_us[f] = fam_us.get(f, 0.0) + dur
            fam_n[f] = fam_n.get(f, 0) + 1
        print(json.dumps({"steps_profiled": nprof, "ms_per_step_timed": elapsed / args.steps * 1e3, "family_ms": {f: v / nprof / 1e3 for f, v in fam_us.items()},
                          "family_launches": {f: n / nprof for f, n in fam_n.items()}, "busy_ms": sum(fam_us.values()) / nprof / 1e3}), flush=True)
        return
    if rank == 0:
        # LoRA: dX only (+ small adapter terms) = 2 x forward; full fine-tune: dX and dW = 3 x forward
        f_step = (3.0 * topology.fwd_flops(cfg, B, h, h, 0)["total"]) if full_ft else 2.0 * topology.fwd_flops(cfg, B, h, h, args.rank)["total"]
        t_step = elapsed / args.steps
        achieved = J * f_step / (ev_ms * 1e-3 / args.steps)
        # HBM-side bytes per step and per-family kernel time: measured around the process (PMC counters and the kernel trace need rocprofv3), handed
        # back through --profile-json; the default workload falls back to the committed profile of this round and says which commit it is from
        traffic = traffic_commit = hbm_kernels = families = traffic_sources_match = None
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        default_workload = (version == "sdxl" and res == 1024 and B == 1 and args.rank == 16 and text is not None and not args.ti_frozen and not full_ft
                            and not args.dora and J == 1)
        import glob
        committed = sorted(glob.glob(os.path.join(pdir, "r[0-9][0-9]_sdxl1024_ti_step_profile.json")))     # the newest round's committed profile
        ppath = args.profile_json or (committed[-1] if (default_workload and committed) else None)
        if ppath and os.path.exists(ppath):
            with open(ppath) as fh:
                prof = json.load(fh)
            traffic, traffic_commit = prof.get("traffic_bytes_per_step"), prof.get("commit")
            # does the quoted profile belong to the code that runs?  (the profile records a hash of the kernel sources + plan code it was taken on)
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("_step_profile_sha", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "step_profile_sha.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                sources_now = mod.kernel_sources_sha()
            except Exception:
                sources_now = None
            traffic_sources_match = (prof.get("kernel_sources_sha16") == sources_now) if (sources_now and prof.get("kernel_sources_sha16")) else None
            # families bound by HBM: algorithmic bytes (every operand once, topology.hbm_bytes) / their kernel time in the profiled step
            alg = topology.hbm_bytes(cfg, B, h, h, args.rank)
            hbm_kernels = {}
            for fam in ("layernorm", "groupnorm", "geglu", "adamw", "lora_grad"):
                ms = prof.get("family_ms", {}).get(fam)
                if ms:
                    moved = prof.get("fetch_bytes_by_family", {}).get(fam, 0.0) + prof.get("write_bytes_by_family", {}).get(fam, 0.0)
                    hbm_kernels[fam] = {"algorithmic_GB": alg[fam] / 1e9, "ms": ms, "GB_per_s": alg[fam] / ms / 1e6, "frac_of_8TBps": alg[fam] / (ms * 1e-3) / 8e12,
                                        "launches": prof.get("family_launches", {}).get(fam), "measured_GB": moved / 1e9 if moved else None}
            # the MFMA-bound families of the profiled step: algorithmic FLOPs (2 x forward census, like `achieved`) / their kernel time, against 2.5 PF
            fl = topology.fwd_flops(cfg, B, h, h, args.rank)
            fam_flop = {"gemm": 2.0 * (fl["total"] - fl["attn_core"] - fl["daam"]), "attention": 2.0 * fl["attn_core"]}
            families = {}
            for fam, fpt in fam_flop.items():
                ms = prof.get("family_ms", {}).get(fam)
                if ms:
                    families[fam] = {"ms": ms, "tflop": fpt / 1e12, "frac": fpt / (ms * 1e-3) / PEAK_BF16_DENSE, "launches": prof.get("family_launches", {}).get(fam)}
        # the per-launch floor model of the step (tools/step_floor.py): what the step's SHAPE allows on this chip
        shape_floor = None
        floors = sorted(glob.glob(os.path.join(pdir, "r[0-9][0-9]_step_floor.json")))
        if default_workload and floors:
            with open(floors[-1]) as fh:
                sf = json.load(fh)
            shape_floor = {"shape_floor_ms": sf.get("shape_floor_ms"), "measured_ms_replayed_per_signature": sf.get("measured_ms"), "calls_per_step": sf.get("calls_per_step"),
                           "commit": sf.get("commit"), "families": sf.get("families"), "largest_gaps": [{k: g[k] for k in ("sig", "calls", "us", "floor_us", "bound", "gap_ms") if k in g} for g in (sf.get("top_gaps") or [])[:8]],
                           "model": (sf.get("model") or {}).get("note"), "source": os.path.basename(floors[-1])}
        if default_workload and not (ppath and os.path.exists(ppath)):
            tpath = os.path.join(pdir, "r02_sdxl1024_ti_hbm_traffic_pmc.json")
            if os.path.exists(tpath):
                with open(tpath) as fh:
                    traffic, traffic_commit = json.load(fh).get("traffic_bytes_per_step"), "round 2 final"
        fpath = os.path.join(pdir, "r01_fullft_hbm_traffic_pmc.json")
        if full_ft and version == "sdxl" and res == 512 and B == 4 and os.path.exists(fpath):
            with open(fpath) as fh:
                traffic, traffic_commit = json.load(fh).get("traffic_bytes_per_step"), "round 1"
        out = {
            "metric": "training images/sec, SDXL 1024px rank-16 LoRA, 1/2/4/8 GPU (job-parallel)",
            "value": world * J * B * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"{version} {res}x{res} FULL-UNet fine-tune batch {B}/GPU: UNet fwd + bwd (dX and every dW), masked/min-SNR MSE, "
                                    f"{'AdamW8bit (block-quantised moments)' if getattr(ts, 'adam8', False) else 'AdamW (fp32 moments)'} over {arena.n / 1e6:.0f} M parameters, bf16 operand refresh" if full_ft else
                                    f"{version} {res}x{res} {'DoRA' if args.dora else 'LoRA'} rank {args.rank} batch {B}" + ("/GPU" if J == 1 else f" per job, {J} concurrent jobs/GPU")
                                    + (": UNet fwd+bwd, masked/min-SNR MSE, AdamW (adapters + magnitudes; per-step weight-norm / scaled-operand refresh)" if args.dora
                                       else ": UNet fwd+bwd, masked/min-SNR MSE, L1, AdamW"))
                                   + (", + textual inversion (text encoders fwd+bwd with 3 trainable tokens, token-attention loss, "
                                      "std regulariser, rows-only AdamW)" + (" [ti lr = 0: frozen-TI fast path, no text-encoder backward]" if args.ti_frozen else "") if text is not None else ", text conditioning injected (--no-ti)"),
                       "text_encoder_fwd_gflop_not_in_roofline": clip_flops / 1e9,
                       "jobs_per_gpu": J, "global_batch": world * J * B,
                       "parallelism": ((f"dp{world}: per bucket {args.ddp_wire or 'fp32'} gradient " + ("reduce-scatter, AdamW on 1/" + str(world) + " of the arena per rank, fp32 all-gather of the masters" if getattr(ts, "zero1", False) else "all-reduce")
                                        + f" ({arena.n * (2 if args.ddp_wire == 'bf16' else 4) / 1e9:.1f} GB of gradients per step, RCCL), overlapped with the weight-gradient GEMMs") if (full_ft and world > 1)
                                       else f"job-parallel x{world * J} ({world} GPU(s) x {J} independent job(s) per GPU, no collective)"
                                            + (f"; a step advances every job once ({J} images per GPU and step), ms_per_step is per such step" if J > 1 else "")),
                       "trained_params": arena.n, "graph": not args.no_graph, "final_loss": loss},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_BF16_DENSE / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_DENSE, "traffic": traffic, "traffic_commit": traffic_commit, "traffic_sources_match": traffic_sources_match,
                         "traffic_GB_per_s": (traffic / (ev_ms * 1e-3 / args.steps) / 1e9) if traffic else None, "hbm_kernels": hbm_kernels, "families": families,
                         "shape_floor_ms": (shape_floor or {}).get("shape_floor_ms"), "shape_floor": shape_floor,
                         "note": f"algorithmic {J} x {f_step / 1e12:.3f} TFLOP per step (2 x fwd census) / {ev_ms / args.steps:.3f} ms "
                                 "per step (HIP events on the replay stream); traffic = HBM-side bytes per step (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, "
                                 f"separate passes) of this command at commit `traffic_commit` (--profile-json, here {os.path.basename(ppath) if ppath else None}: the newest committed profiles/rNN_sdxl1024_ti_step_profile.json by default); "
                                 "hbm_kernels: the HBM-bound kernel families - algorithmic bytes (topology.hbm_bytes) / their time in the profiled step, against 8 TB/s; "
                                 "families: the MFMA-bound kernel families of the same profiled step - algorithmic TFLOP (2 x fwd census) / their kernel time, against 2.5 PFLOP/s"},
        }
        out["telemetry"] = dict(telemetry or {}, note="shader clock (mean over the XCDs) / socket power / hotspot temperature of this GPU sampled every 20 ms from a host thread "
                                "over the timed region - the chip clocks to its power budget, so the same kernels time differently on a cold and a warm box (DESIGN 6)")
        if world == 1 and not args.no_sustained and not args.no_graph:
            # Extra object (never `value`): the same replay loop for `--sustained-steps` more steps right behind the timed region - the figure a job sees once the
            # chip has settled into its power budget, with its own clock / power samples; the K-step figure above is what the contract asks for
            tele2 = GpuTelemetry(index=0).start()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.sustained_steps):
                step_all(args.warmup + (i % args.steps))
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
            out["sustained"] = {"steps": args.sustained_steps, "ms_per_step": dts / args.sustained_steps * 1e3, "value": J * B * args.sustained_steps / dts, "unit": "images/s",
                                "telemetry": tele2.stop(), "note": "extra measurement, not `value`: the same graph replays continued for more steps (host clock around a synchronize pair)"}
        if dry:
            # one more step with the collectives recorded: what crosses the wire per step and what a ring over xGMI needs for it.  Ring reduce-scatter / all-gather move
            # (N - 1) / N of the buffer through each GPU's slowest link, an all-reduce twice that; one xGMI link of an MI355X carries ~153 GB/s per direction (the prompt's
            # figure; 7 links per GPU, but a ring uses one neighbour link each way).  Lower bound of the exposed time: the collectives overlap the weight-gradient GEMMs.
            ts.coll_log = []
            ts.run(lr_at(total, total), 0.0)
            torch.cuda.synchronize()
            log, ts.coll_log = ts.coll_log, None
            link = 153e9
            by_op = {}
            for op, _, n_in, _, n_out, dt, _ in log:
                e = by_op.setdefault(op, {"calls": 0, "bytes": 0})
                e["calls"] += 1
                e["bytes"] += max(n_in, n_out) * (2 if dt == "bfloat16" else 4)
            wire = {}
            for n in (2, 4, 8):
                vol = sum(e["bytes"] * (n - 1) / n * (2 if op.startswith("all_reduce") else 1) for op, e in by_op.items())      # bytes every GPU sends (and receives) per step
                # one ring = one neighbour link; the node is fully connected (a link to each of the other N - 1 GPUs), so a direct exchange spreads the same volume over N - 1 links
                wire[str(n)] = {"sent_GB_per_gpu": vol / 1e9, "one_ring_ms": vol / link * 1e3, "all_links_direct_ms": vol / (link * (n - 1)) * 1e3}
            out["collectives"] = {"sequence": [f"{op} x{e['calls']}: {e['bytes'] / 1e9:.3f} GB" for op, e in by_op.items()], "calls_per_step": len(log),
                                  "all_views_16B_aligned": all(e[6] for e in log), "wire_time_per_step_by_gpus": wire,
                                  "note": "1-rank RCCL group (--dry-collectives): the call sequence, buffers and graph phases of --gpus N, executed; wire time = bytes x (N-1)/N "
                                          "(x2 for all-reduce) / 153 GB/s per xGMI link, over one link (a single ring) and over all N-1 links of the fully connected node - bounds "
                                          "that the overlap with the weight-gradient GEMMs hides or not"}
            out["config"]["parallelism"] = (f"dry run of dp-N on one GPU: {len(log)} collectives per step on a 1-rank RCCL communicator (" + "; ".join(out["collectives"]["sequence"]) + ")")
        print(f"[bench] timed region done: {t_step * 1e3:.2f} ms/step; extras follow", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline and not full_ft:
            # measured in a CHILD process with a wall-clock limit: nothing on the host side may keep the JSON line from being printed
            child = run_guarded(["--cpu-baseline-only", "--config", version, "--res", str(res), "--rank", str(args.rank)] + (["--no-ti"] if args.no_ti else []), 420)
            if child:
                cb = child[-1]                  # the last complete measurement (full size if it finished, else the calibration sample)
                dt = sum(cb["times"]) / len(cb["times"])
                scaled = dt * (f_step / B) / cb["flops"]     # seconds per full-size image on this host (== dt when the sample IS the full size)
                same = cb["hw"] == h
                out["cpu_baseline"] = {"value": 1.0 / scaled, "unit": "images/s", "cores": cb["cores"], "kind": "port",
                                       "step_seconds": [round(x, 3) for x in cb["times"]],
                                       "sample": f"fp32 oracle (CPU port of the reference path, oracle.step_ref.RefTrainer.step: {cb.get('what', 'UNet fwd+bwd')} - the step the GPU line times) at {cb['hw'] * 8}x{cb['hw'] * 8} B=1 after a warm-up: "
                                                 + ", ".join(f"{x:.2f} s" for x in cb["times"]) + f" ({cb['flops'] / 1e12:.2f} TFLOP each, {cb['cores']} threads)"
                                                 + ("" if same else f"; scaled by the FLOP ratio to the {res}x{res} workload (the full size did not fit the time budget of the default run)")}
            print("[bench] cpu baseline done", file=sys.stderr, flush=True)
            if not child:
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": usable_cores(), "kind": "port", "step_seconds": [],
                                       "sample": "the CPU oracle did not complete a step within the 420 s guard of the default run"}
        if world == 1 and J == 1 and not full_ft and not args.no_cpu_baseline and not args.no_library_gpu and text is not None and not args.no_ti and not args.dora and version in ("sdxl", "sd15") and not args.batch:
            # Extra object beside cpu_baseline (never `value`): the same torch restatement of the step on THIS GPU through PyTorch-ROCm's library kernels
            # (library_gpu_baseline), in a child process with a wall-clock limit; the parent's model stays resident, the child needs ~25 GB more
            if B == 1:
                child = run_guarded(["--library-gpu-only", "--config", version, "--res", str(res), "--rank", str(args.rank)], 300)
                if child:
                    out["library_gpu_baseline"] = child[-1]
                    out["library_gpu_baseline"]["speedup_of_value"] = out["value"] / child[-1]["value"]
                print("[bench] library GPU baseline done", file=sys.stderr, flush=True)
        if world == 1 and J == 1 and not args.no_graph and not args.no_families and not args.no_cpu_baseline:
            # Extra measurement (never `value`), child process with a wall-clock limit: the per-family kernel times of THIS box and THIS code (torch.profiler over three replayed steps of
            # the same workload) next to the families of the committed rocprofv3 profile above; fractions against the same algorithmic FLOPs
            argv = ["--families-only", "--config", version, "--res", str(res), "--rank", str(args.rank), "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--no-sustained"] \
                + (["--batch", str(args.batch)] if args.batch else []) + (["--no-ti"] if args.no_ti else []) + (["--ti-frozen"] if args.ti_frozen else []) + (["--dora"] if args.dora else []) \
                + (["--full-ft"] if full_ft else []) + (["--fp32-moments"] if args.fp32_moments else [])
            child = run_guarded(argv, 300)
            if child and isinstance(child[-1], dict) and "family_ms" in child[-1]:
                fr = child[-1]
                if not full_ft:
                    fl2 = topology.fwd_flops(cfg, B, h, h, args.rank)
                    for fam, fpt in (("gemm", 2.0 * (fl2["total"] - fl2["attn_core"] - fl2["daam"])), ("attention", 2.0 * fl2["attn_core"])):
                        if fr["family_ms"].get(fam):
                            fr.setdefault("frac_of_peak", {})[fam] = fpt / (fr["family_ms"][fam] * 1e-3) / PEAK_BF16_DENSE
                fr["note"] = "torch.profiler (roctracer kernel records) over three replayed steps in a child process of this run: kernel time per family and step on THIS box with THIS code"
                out["roofline"]["families_this_run"] = fr
            print("[bench] in-run family times done", file=sys.stderr, flush=True)
        if world == 1 and J == 1 and not full_ft and not args.no_graph and not args.no_train_loop and text is not None:
            # Extra measurement (never `value`), in a child process with a wall-clock limit (see train_loop_measure)
            child = run_guarded(["--train-loop-only", "--config", version, "--res", str(res), "--rank", str(args.rank), "--steps", str(args.steps)]
                                + (["--batch", str(args.batch)] if args.batch else []), 420)
            if child:
                out["train_loop"] = child[-1]
            print("[bench] train loop done", file=sys.stderr, flush=True)
        if world == 1 and J == 1 and not full_ft and not args.no_graph and not args.no_concurrent:
            # Extra measurement (never `value`): the same workload with TWO independent jobs stepped concurrently on this GPU,
            # each on its own stream with its own hipGraph (train.train_concurrent).
            from sd_lora_trainer_amd import ops as _ops
            _ops.set_throughput_hint(True)
            try:
                st2 = [torch.cuda.Stream(device=device) for _ in range(2)]
                pair = []
                for j, st in enumerate(st2):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        pair.append(build_job(8 + j)[0])
                    st.synchronize()

                def pair_step(i):
                    for tsj, st in zip(pair, st2):
                        with torch.cuda.stream(st):
                            tsj.run(lr_at(i, total), ti_lr * (1 - i / total) ** 1.7)
                for i in range(args.warmup):
                    pair_step(i)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    pair_step(args.warmup + i)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                assert all(math.isfinite(tsj.total_loss()) for tsj in pair)
                out["two_jobs_per_gpu"] = {"value": 2 * B * args.steps / dt2, "unit": "images/s", "ms_per_pair_of_steps": dt2 / args.steps * 1e3,
                                           "note": "extra measurement, not `value`: two independent jobs of the same workload in this process, one stream + "
                                                   "hipGraph each (python bench.py --jobs-per-gpu 2 reports it as the main figure)"}
            finally:
                _ops.set_throughput_hint(False)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
