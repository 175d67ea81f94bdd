"""Job-parallel multi-GPU path on CPU: device pinning of the sweep launcher and the timing
protocol of bench.py (barrier + MAX over ranks) with world_size 2 on the gloo backend."""
import os
import socket
import time

import torch
import torch.multiprocessing as mp

from sd_lora_trainer_amd import parallel


def test_job_pinning_and_sweep_plan():
    cfgs = [f"cfg_{i}.json" for i in range(11)]
    plan = parallel.sweep_plan(cfgs, 8)
    assert [g for _, g, _ in plan] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2]
    assert [w for _, _, w in plan] == [0] * 8 + [1] * 3
    launched = parallel.run_sweep(cfgs, 8, dry_run=True)
    assert len(launched) == 11
    for i, (cmd, env) in enumerate(launched):
        assert cmd[-1] == cfgs[i] and env["HIP_VISIBLE_DEVICES"] == str(i % 8) == env["CUDA_VISIBLE_DEVICES"]
    e = parallel.job_env(5, 4, base_env={})
    assert e["HIP_VISIBLE_DEVICES"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    parallel.barrier_sync()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # rank-dependent "step" time: the slowest replica defines the job
    parallel.barrier_sync()
    elapsed = 0.05 * (rank + 1)
    mx = parallel.max_over_ranks(elapsed)
    thr = parallel.aggregate_throughput(10, elapsed)
    wall = time.perf_counter() - t0
    out.put((rank, mx, thr, wall))
    torch.distributed.destroy_process_group()


def test_timing_protocol_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mx, thr, wall in res:
        assert abs(mx - 0.10) < 1e-9                 # MAX over ranks
        assert abs(thr - 2 * 10 / 0.10) < 1e-6       # whole-job throughput = sum of units / slowest replica
        assert wall >= 0.10 - 1e-3                   # the barrier made the fast rank wait for the slow one


# ------------------------------------------------------------------------------------------------ data-parallel full fine-tune
def _ddp_worker(rank, world, port, out, wire="fp32", zero1=True, optimizer="adamw"):
    """Each rank holds ONE sample of a 2-sample batch; after the gradient all-reduce (sum, mean folded into the optimizer's
    gradient scale) both ranks must hold the parameters a single process gets from the full batch."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import unet as unet_mod
    from tests import emu_ops
    from tests.test_fullft_cpu import _inputs
    parallel.init_distributed("gloo")
    cfg, h = U.CONFIGS["tiny15"], 16
    sd = U.init_unet_state(cfg, seed=0)
    latent, noise, mask, t, ctx, _, _, _ = _inputs(cfg, 2, h)
    mask = torch.ones_like(mask)        # per-sample losses independent of the rest of the batch (gamma = 0 branch normalises by the batch mask mean)
    rt = unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
    tr = fullft.WeightTrainer(rt)
    tr.bucket_floats = 150_000          # several buckets on the toy model: weight gradients + all-reduce bucket by bucket
    unet = unet_mod.UNet(rt, topology.CONFIGS["tiny15"], sd, trainer=tr)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=0.0, process_group=True, ddp_wire_dtype=wire, ddp_zero1=zero1, optimizer=optimizer)
    assert (ts.wire is not None) == (wire == "bf16") and ts.zero1 == (zero1 and optimizer in ("adamw", "AdamW8bit"))
    assert ts.adam8 == (optimizer == "AdamW8bit") and (tr.q8_sh is not None) == (ts.adam8 and ts.zero1)
    zero1 = ts.zero1
    if zero1:          # the moments exist for the owned slices only; every bucket splits evenly
        assert tr.m is None and tr.m_sh.numel() * world == tr.n_mat and all((o1 - o0) % (4 * world) == 0 for o0, o1 in tr.buckets)
    assert ts.bucketed and len(tr.buckets) >= 4 and tr.buckets[0][0] == 0 and tr.buckets[-1][1] == tr.n_mat
    assert all(a[1] == b[0] for a, b in zip(tr.buckets, tr.buckets[1:]))          # contiguous cover of the matrix region
    s = slice(rank, rank + 1)
    for it in range(2):
        ts.set_batch(latent[s], noise[s], t[s], mask[s], ctx[s])
        ts.coll_log = [] if it == 1 else None       # the second step's collectives, in call order
        ts.run(1e-3)
    own = [tr.shard_range(b) for b in range(len(tr.buckets))] if zero1 else None
    gn = ts.grad_norm()                 # (a collective under ZeRO-1: every rank calls it)
    extra = dict(state=ts.prodigy.state.numpy().copy(), s=ts.prodigy.s.numpy().copy()) if ts.prodigy is not None else None
    out.put((rank, tr.params.numpy().copy(), float(ts.loss), tr.grads.numpy().copy(), own, tr.n_mat, extra, gn, (ts.coll_log, list(tr.buckets), tr.n)))      # numpy: pickled through the pipe (a shared-memory tensor dies with the worker)
    torch.distributed.destroy_process_group()


import pytest


def _run_ddp(wire, zero1, optimizer="adamw"):
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = _free_port()
    procs = [ctx_mp.Process(target=_ddp_worker, args=(r, 2, port, q, wire, zero1, optimizer)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_fullft_zero1_equals_allreduce_two_ranks_gloo():
    """The sharded exchange (reduce-scatter -> AdamW on 1 / world of every bucket -> all-gather of the masters -> operand refresh) leaves the
    SAME BITS as all-reduce + the full AdamW on every rank: with two ranks the sum a + b has one order, AdamW is element-wise, and the
    masters of a slice are computed once and copied.  (With more ranks the two collectives may order their sums differently.)"""
    z, a = _run_ddp("fp32", True), _run_ddp("fp32", False)
    assert torch.equal(torch.from_numpy(z[0][1]), torch.from_numpy(z[1][1])), "ZeRO-1 replicas diverged"
    assert torch.equal(torch.from_numpy(z[0][1]), torch.from_numpy(a[0][1])), "sharded optimizer != replicated optimizer"
    assert z[0][2] == a[0][2] and z[1][2] == a[1][2]


def test_fullft_zero1_adamw8bit_two_ranks_gloo():
    """`AdamW8bit` with the optimizer sharded (enable_zero1(adam8=True): byte moments + absmax per 2048 consecutive elements of the owned slices, sdlt_adamw8_flat): the
    replicas stay bit-identical (every master element is computed by one rank and copied), and after the two steps the masters are within a few percent of the
    fp32-moment run's displacement (the first step uses unquantised moments: identical up to the order of decay and step)."""
    q, z = _run_ddp("fp32", True, "AdamW8bit"), _run_ddp("fp32", True)
    p0, p1, pz = torch.from_numpy(q[0][1]), torch.from_numpy(q[1][1]), torch.from_numpy(z[0][1])
    assert torch.equal(p0, p1), "AdamW8bit ZeRO-1 replicas diverged"
    assert not torch.equal(p0, pz)
    nm = q[0][5]
    dev = (p0 - pz).abs()
    # two steps at lr 1e-3: on average the second step's quantised moments move an element by a tenth of lr against the fp32-moment run, and all but a 1e-3 fraction by
    # less than 2 lr.  The tail is a property of the restated algorithm, not of the kernel: an element whose gradient is ~1e-7 of its block's largest has its second moment
    # rounded to 0 and - when negative - its first moment rounded AWAY from 0 (bitsandbytes' sign rule: the code of 0 counts as positive), so its next update divides an
    # inflated m by ~eps (oracle/adam8bit_ref.py says the same).  The vector region keeps fp32 moments: equal to the fp32 run up to the order of decay and step.
    assert float(dev[:nm].mean()) <= 1.5e-4 and float((dev[:nm] > 2e-3).float().mean()) <= 1e-3, (float(dev[:nm].mean()), float((dev[:nm] > 2e-3).float().mean()))
    assert float(dev[nm:].max()) <= 1e-5          # (its second-step gradients saw slightly different matrices)


def test_fullft_data_parallel_prodigy_two_ranks_gloo():
    """Prodigy under data parallelism (refused until round 4): the summed gradients become the mean before Prodigy's two passes, so both ranks
    hold the state ONE process gets from the whole batch - replicas bit-identical, parameters equal to the single-process run's up to the
    rounding of (g0 + g1) / 2 against the batch-mean gradient."""
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import unet as unet_mod
    from tests import emu_ops
    from tests.test_fullft_cpu import _inputs
    res = _run_ddp("fp32", True, "prodigy")
    p0, p1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(p0, p1), "ranks diverged"
    cfg, h = U.CONFIGS["tiny15"], 16
    sd = U.init_unet_state(cfg, seed=0)
    latent, noise, mask, t, ctx, _, _, _ = _inputs(cfg, 2, h)
    mask = torch.ones_like(mask)
    rt = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    tr = fullft.WeightTrainer(rt)
    unet = unet_mod.UNet(rt, topology.CONFIGS["tiny15"], sd, trainer=tr)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=0.0, optimizer="prodigy")
    for _ in range(2):
        ts.set_batch(latent, noise, t, mask, ctx)
        ts.run(1e-3)
    # Prodigy's state is what a wrong gradient scale would corrupt (at d0 = 1e-6 two steps move the parameters by ulps): the scalars
    # d, d_max, d_numerator, d_denom, d_hat and the per-element s = sum of d-weighted gradients against ONE process on the whole batch
    st_d, st_1 = torch.from_numpy(res[0][6]["state"]), ts.prodigy.state
    assert torch.equal(st_d, torch.from_numpy(res[1][6]["state"]))
    assert float(st_1[3].abs()) > 0 and float(st_1[4]) > 0, st_1
    torch.testing.assert_close(st_d[:7], st_1[:7], rtol=2e-3, atol=0)
    s_d, s_1 = torch.from_numpy(res[0][6]["s"]), ts.prodigy.s
    assert float((s_d - s_1).norm() / s_1.norm()) <= 2e-3
    # with the SUM instead of the mean the numerator would be 2x, the denominator 2x: d_hat equal - but s itself 2x off
    assert float((2 * s_d - s_1).norm() / s_1.norm()) > 0.5


@pytest.mark.parametrize("wire,zero1", [("fp32", True), ("bf16", True), ("fp32", False), ("bf16", False)])
def test_fullft_data_parallel_two_ranks_gloo(wire, zero1):
    """zero1 (the default): reduce-scatter + sharded AdamW + all-gather; zero1 = False: all-reduce + replicated AdamW.
    wire = bf16: the matrix gradients are packed to bf16 per bucket before the exchange and unpacked after it (TrainStep(ddp_wire_dtype=)):
    replicas still bit-identical, the reduced gradient equals the full-batch gradient to bf16 precision."""
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import unet as unet_mod
    from tests import emu_ops
    from tests.test_fullft_cpu import _inputs
    res = _run_ddp(wire, zero1)
    # single process, both samples in one batch
    cfg, h = U.CONFIGS["tiny15"], 16
    sd = U.init_unet_state(cfg, seed=0)
    latent, noise, mask, t, ctx, _, _, _ = _inputs(cfg, 2, h)
    mask = torch.ones_like(mask)
    rt = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    tr = fullft.WeightTrainer(rt)
    unet = unet_mod.UNet(rt, topology.CONFIGS["tiny15"], sd, trainer=tr)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=0.0)
    for _ in range(2):
        ts.set_batch(latent, noise, t, mask, ctx)
        ts.run(1e-3)
    p0, p1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(p0, p1), "ranks diverged"
    # the all-reduced arena holds the SUM of the ranks' gradients = 2 x the gradient of the batch-mean loss
    g_ddp, g_one = torch.from_numpy(res[0][3]).clone(), tr.grads
    if zero1:           # after the reduce-scatter a rank holds the sums of its own slices only: assemble the reduced arena from the owners
        for r in range(2):
            for s0, s1 in res[r][4]:
                g_ddp[s0:s1] = torch.from_numpy(res[r][3])[s0:s1]
        assert sum(s1 - s0 for r in range(2) for s0, s1 in res[r][4]) == res[0][5]          # the slices of the two ranks tile the matrix region
    g_ddp = g_ddp / 2
    # the debug read-out (main.py:373-379) is the norm of the global batch's gradient on every rank, sharded exchange or not
    assert res[0][7] == res[1][7] or not zero1
    for r in range(2):
        ref_norm = float(g_ddp.double().norm())
        # all-reduce path: torch's fp32 norm of the whole arena (its own summation order); sharded path: fp64 partial sums
        assert abs(res[r][7] - ref_norm) <= (1e-6 if zero1 else 2e-3) * ref_norm, (res[r][7], ref_norm)
    if wire == "bf16":      # two bf16 roundings (the pack, the sum over the ranks) of each rank's share: 2^-8 of the addends
        assert float((g_ddp - g_one).norm() / g_one.norm()) <= 6e-3
        assert float((g_ddp - g_one).abs().max()) <= 1e-2 * float(g_one.abs().max())
    else:
        assert float((g_ddp - g_one).abs().max()) <= 2e-3 * float(g_one.abs().max())     # 2nd step: the replicas already differ by the sign-noise above
    # parameters: identical up to Adam's sign(g) steps where the gradient is analytically zero (e.g. a conv bias in front of
    # a GroupNorm: +-1e-9 of rounding noise becomes +-lr), so only a small fraction of the elements may differ
    frac = float(((p0 - tr.params).abs() > 5e-2 * 2e-3).float().mean())
    assert frac < 0.02, frac
    assert abs(0.5 * (res[0][2] + res[1][2]) - float(ts.loss)) <= 1e-4 * abs(float(ts.loss))


# ------------------------------------------------------------------------------------------------ train() under data parallelism
def _train_ddp_worker(rank, world, port, out, tmp, te_lora=False):
    """The whole train() generator (main.py:34-551 mirror) as a 2-rank data-parallel full fine-tune on a dataset whose size is NOT a
    multiple of the world size: every rank must run the same number of steps (each step issues collectives, the checkpoint a barrier)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir(tmp)
    torch.set_num_threads(2)
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import train as T
    from sd_lora_trainer_amd import unet as unet_mod
    from sd_lora_trainer_amd.config import TrainingConfig
    from tests import emu_ops
    parallel.init_distributed("gloo")
    cfg = TrainingConfig(lora_training_urls="synthetic:5", concept_mode="object", pretrained_model={"path": "synthetic:tiny15"}, seed=3, resolution=128,
                         train_batch_size=1, max_train_steps=7, is_lora=False, unet_optimizer_type="adamw", unet_lr=1e-4, ti_lr=1e-3,
                         n_sample_imgs=0, checkpointing_steps=1000, output_dir=os.path.join(tmp, f"out_rank{rank}"),
                         **(dict(text_encoder_lora_optimizer="adamw", text_encoder_lora_lr=1e-3, text_encoder_lora_rank=4) if te_lora else {}))
    n_calls = [0]
    real = step_mod.TrainStep._run

    def counted(self, *a, **k):
        n_calls[0] += 1
        return real(self, *a, **k)
    step_mod.TrainStep._run = counted
    rt = unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
    gen = T.train(cfg, runtime=rt)
    holder = {}
    real_init = step_mod.TrainStep.__init__

    e_init = {}

    def spy_init(self, *a, **k):
        real_init(self, *a, **k)
        holder["ts"] = self
        e_init["conv_in"] = self.group.view(self.group.by_name["conv_in.weight"]).clone()
    step_mod.TrainStep.__init__ = spy_init
    try:
        while True:
            next(gen)
    except StopIteration as e:
        config, _ = e.value
    ts = holder["ts"]
    # (the weights a frozen-TI step must still train: a matrix weight's master after the run vs its checkpoint value)
    e = ts.group.by_name["conv_in.weight"]
    moved = float((ts.group.view(e) - e_init["conv_in"]).abs().max()) if "conv_in" in e_init else -1.0
    te = None
    if ts.te_arena is not None:
        te = (ts.te_arena.params.numpy().copy(), float(ts.te_arena.m.abs().max()))
    out.put((rank, n_calls[0], config.num_train_epochs, ts.group.params.numpy().copy(), ts.ti.params.numpy().copy(), moved, te))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("te_lora", [False, True])
def test_train_data_parallel_uneven_dataset_two_ranks_gloo(tmp_path, te_lora):
    """te_lora: text-encoder LoRA next to the data-parallel full fine-tune (refused until round 4): its adapter gradients are exchanged like
    the token rows, so the adapters stay identical on all ranks and train."""
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = _free_port()
    procs = [ctx_mp.Process(target=_train_ddp_worker, args=(r, 2, port, q, str(tmp_path), te_lora)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, n0, ep0, p0, t0, mv0, te0), (_, n1, ep1, p1, t1, mv1, te1) = res
    if te_lora:
        assert te0 is not None and (te0[0] == te1[0]).all(), "text-encoder adapters diverged"
        assert te0[1] > 0, "the text-encoder adapters received no gradient"
    assert mv0 > 0 and mv1 > 0, "the matrix weights did not move: the weight-gradient flush / exchange of the step was skipped"
    # 5 images over 2 ranks: 3 per rank and epoch (the shuffle wraps around), max_train_steps + 1 = 8 optimizer steps on BOTH ranks
    assert n0 == n1 == 8 and ep0 == ep1 == 3
    assert (p0 == p1).all(), "UNet replicas diverged"
    assert (t0 == t1).all(), "token rows diverged (their gradients are exchanged too: the ranks see different captions)"


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` outside a launcher must start 2 ranks itself (round 2 parsed the flag and ran one job): the line
    says n_gpus 2 and names 2 ranks; `--gpus 2` on a node without 2 GPUs refuses instead of printing a 1-GPU number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-test"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and "x2" in lines[0]["config"]["parallelism"] and "2 rank" in lines[0]["config"]["parallelism"]
    assert lines[0]["ms_per_step"] >= 4.0 - 0.5              # MAX over ranks: the slow rank (4 ms sleeps) defines the step
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _check_collective_sequence(log, buckets, n, n_mat, rank, world, zero1, wire):
    """The exchange step as RCCL will see it (step.TrainStep.flush_and_reduce / gather_params): per bucket, in flush order, one in-place reduce-scatter whose output is
    THIS rank's slice of its input (output = input + rank * count: the form NCCL / RCCL runs without a staging copy) - or one all-reduce -, then the replicated
    vector region, then (ZeRO-1) one in-place all-gather per bucket whose input is this rank's slice of the output; every view 16-byte aligned."""
    dt = "bfloat16" if wire == "bf16" else "float32"
    ops = [e[0] for e in log]
    nb = len(buckets)
    if zero1:
        assert ops == ["reduce_scatter"] * nb + ["all_reduce:vec"] + ["all_gather"] * nb, ops
    else:
        assert ops == ["all_reduce"] * nb + ["all_reduce:vec"], ops
    for b, (o0, o1) in enumerate(buckets):
        op, i_off, i_n, o_off, o_n, dtype, aligned = log[b]
        assert aligned and dtype == dt and i_off == o0, (log[b], buckets[b])
        if zero1:
            cnt = (o1 - o0) // world
            assert i_n == o1 - o0 and cnt * world == i_n and o_n == cnt and o_off == i_off + rank * cnt, (log[b], buckets[b], rank)
            ag = log[nb + 1 + b]
            assert ag[0] == "all_gather" and ag[5] == "float32" and ag[6] and ag[3] == o0 and ag[4] == o1 - o0 and ag[2] == cnt and ag[1] == o0 + rank * cnt, (ag, buckets[b])
        else:
            assert i_n == o_n and o_off == i_off and i_n == min(o1, n_mat) - o0
    vec = log[nb]
    assert vec[1] == n_mat and vec[2] == n - n_mat and vec[5] == "float32"
    assert buckets[0][0] == 0 and buckets[-1][1] == n_mat and all(a[1] == b_[0] for a, b_ in zip(buckets, buckets[1:]))


@pytest.mark.parametrize("wire,zero1", [("fp32", True), ("bf16", True), ("fp32", False)])
def test_ddp_collective_call_sequence_two_ranks_gloo(wire, zero1):
    """What the first RCCL run of `bench.py --full-ft --gpus N` will execute, asserted on two gloo ranks: call order, slice arithmetic and alignment of every collective
    of one optimizer step (make the first multi-GPU run a measurement, not a debugging session)."""
    res = _run_ddp(wire, zero1)
    for r in range(2):
        log, buckets, n = res[r][8]
        _check_collective_sequence(log, buckets, n, res[r][5], r, 2, zero1, wire)
    assert [e[:6] for e in res[0][8][0] if not e[0].startswith("reduce_scatter") and not e[0].startswith("all_gather")] == \
           [e[:6] for e in res[1][8][0] if not e[0].startswith("reduce_scatter") and not e[0].startswith("all_gather")]      # rank-independent collectives are issued identically


def _forced_worker(port, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import torch.distributed as dist
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import unet as unet_mod
    from tests import emu_ops
    from tests.test_fullft_cpu import _inputs
    dist.init_process_group("gloo", rank=0, world_size=1)
    cfg, h = U.CONFIGS["tiny15"], 16
    sd = U.init_unet_state(cfg, seed=0)
    latent, noise, mask, t, ctx, _, _, _ = _inputs(cfg, 1, h)

    def build(ddp):
        rt = unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
        tr = fullft.WeightTrainer(rt)
        tr.bucket_floats = 150_000
        unet = unet_mod.UNet(rt, topology.CONFIGS["tiny15"], {k: v.clone() for k, v in sd.items()}, trainer=tr)      # (a CPU runtime aliases the tensors it is given)
        ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=0.0, **(dict(process_group=True, ddp_force=True) if ddp else {}))
        return ts, tr
    ts, tr = build(True)
    assert ts.ddp and ts.bucketed and ts.zero1 and ts.world == 1
    ts1, tr1 = build(False)
    for it in range(2):
        for x in (ts, ts1):
            x.set_batch(latent, noise, t, mask, ctx)
            x.coll_log = [] if (it == 1 and x is ts) else None
            x.run(1e-3)
    _check_collective_sequence(ts.coll_log, list(tr.buckets), tr.n, tr.n_mat, 0, 1, True, "fp32")
    out.put((torch.equal(tr.params, tr1.params), float(ts.loss), float(ts1.loss), len(ts.coll_log)))
    dist.destroy_process_group()


def test_ddp_forced_on_one_rank_runs_the_exchange_step_and_changes_nothing():
    """ddp_force (bench.py --dry-collectives): the data-parallel exchange step on a ONE-rank group - same call sequence as with N ranks (checked), and the training
    result is bit-identical to the plain single-process step (a 1-rank reduce-scatter / all-gather is the identity; AdamW on '1 / 1 of every bucket' is AdamW)."""
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    p = ctx_mp.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    same, l_ddp, l_one, ncoll = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and l_ddp == l_one and ncoll >= 9, (same, l_ddp, l_one, ncoll)
