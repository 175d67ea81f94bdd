"""Data-parallel full fine-tune on the MI355X: the bucketed, overlapped gradient exchange of step.TrainStep (forward+backward graph,
one hipGraph per weight-gradient bucket, asynchronous all-reduce per bucket, optimizer graph) with TWO ranks.  gpurun gives one
GPU, so both ranks share cuda:0 and the collective is gloo (host-staged) instead of RCCL - the graph / bucket / async-work plumbing
is the same code the 8-GPU run uses with backend nccl.  Checks: replicas bit-identical after several steps, loss finite and falling."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, zero1=True):
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import unet as unet_mod
    from tests.test_fullft_cpu import _inputs
    cfg, h = U.CONFIGS["tinyxl"], 16
    sd = U.init_unet_state(cfg, seed=0)
    latent, noise, mask, t, ctx, pooled, tid, _ = _inputs(cfg, 2, h)
    rt = unet_mod.Runtime("cuda:0", 1)
    tr = fullft.WeightTrainer(rt)
    tr.bucket_floats = 150_000
    unet = unet_mod.UNet(rt, topology.CONFIGS["tinyxl"], sd, trainer=tr)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, process_group=True, ddp_zero1=zero1)
    assert ts.bucketed and len(tr.buckets) >= 3 and ts.zero1 == zero1
    s = slice(rank, rank + 1)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent[s]), dv(noise[s]), dv(t[s]), dv(mask[s]), dv(ctx[s]), dv(pooled[s]), dv(tid[s]))
    ts.capture(warmup=1)
    # forward+backward | one per bucket | optimizer  (ZeRO-1: sharded AdamW | all-gather outside the graphs | operand refresh)
    assert len(ts.graphs) == len(tr.buckets) + (3 if zero1 else 2)
    losses = []
    for i in range(6):
        ts.run(2e-4)
        losses.append(float(ts.loss))
    torch.cuda.synchronize()
    return (rank, tr.params.cpu().numpy().copy(), losses)


SCENARIOS = [("plain", True), ("plain", False), ("ti", True), ("ti", False)]


def _rank_main(rank, world, port, out, threads):
    """One process per rank for the whole module: the four scenarios run back to back on one gloo group (round 5 spawned a fresh pair of processes -
    torch import, HIP context, 256 default OpenMP threads on a 16-CPU quota - for each: 290 s of the suite)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import traceback
    import torch.distributed as dist
    torch.set_num_threads(threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    for kind, zero1 in SCENARIOS:
        try:
            res = (_worker if kind == "plain" else _worker_ti)(rank, world, zero1)
            out.put((kind, zero1, rank, res, None))
        except Exception:                     # reported per scenario; the other rank would hang in its next collective, so this rank stops here
            out.put((kind, zero1, rank, None, traceback.format_exc()))
            break
        dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def ddp_results():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.conftest import _usable_cores
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = _free_port()
    procs = [ctx_mp.Process(target=_rank_main, args=(r, 2, port, q, max(1, _usable_cores() // 2))) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        while len(got) < 2 * len(SCENARIOS):
            kind, zero1, rank, res, err = q.get(timeout=600)
            got[(kind, zero1, rank)] = (res, err)
            if err is not None:
                break
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return got


def _scenario(ddp_results, kind, zero1):
    out = []
    for rank in range(2):
        assert (kind, zero1, rank) in ddp_results, f"rank {rank} never reported scenario {kind} zero1={zero1}: {[v[1] for v in ddp_results.values() if v[1]]}"
        res, err = ddp_results[(kind, zero1, rank)]
        assert err is None, err
        out.append(res)
    return out


@pytest.mark.parametrize("zero1", [True, False])
def test_bucketed_ddp_two_ranks_on_one_gpu(ddp_results, zero1):
    """zero1 (default): reduce-scatter per bucket -> AdamW on the owned slices -> all-gather of the masters -> operand refresh;
    zero1 = False: all-reduce per bucket -> the full AdamW on every rank."""
    res = _scenario(ddp_results, "plain", zero1)
    p0, p1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(p0, p1), "data-parallel replicas diverged"
    for _, _, losses in res:
        assert all(x == x for x in losses)
    mean = [0.5 * (a + b) for a, b in zip(res[0][2], res[1][2])]
    assert mean[-1] < mean[0], mean


def _worker_ti(rank, world, zero1=True):
    """Full fine-tune + textual inversion under data parallelism, CAPTURED (the default workload of full_finetuning_example.json: ti_lr > 0).
    Round 4 crashed here inside torch.cuda.graph: the frozen-TI graph variants were captured also for world > 1 and recorded the full-arena
    AdamW, whose moments ZeRO-1 has released."""
    import sd_lora_trainer_amd.clip as clip_mod
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    from sd_lora_trainer_amd import step as step_mod
    from sd_lora_trainer_amd import unet as unet_mod
    from tests.test_ti_step_cpu import EOS, NTOK, _captions, _hf
    cfg, h = U.CONFIGS["tinyxl"], 32
    sd = U.init_unet_state(cfg, seed=0)
    hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(2, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(2, 4, h, h, generator=g)
    mask = torch.ones(2, 4, h, h)
    t = torch.tensor([10, 900])
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * 2)
    lists, ids = _captions(2)
    rt = unet_mod.Runtime("cuda:0", 1)
    tr = fullft.WeightTrainer(rt)
    tr.bucket_floats = 150_000
    unet = unet_mod.UNet(rt, topology.CONFIGS["tinyxl"], sd, trainer=tr)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=1, act="quick_gelu", mode="penultimate", with_projection=False, n_train=NTOK),
            clip_mod.ClipTextEncoder(rt, "te2", sds[1], heads=1, act="gelu", mode="penultimate", with_projection=True, n_train=NTOK)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, text=text, n_tokens=NTOK, token_attention_loss_w=2e-2, ti_std_loss_w=0.01,
                            process_group=True, ddp_zero1=zero1)
    assert ts.zero1 == zero1 and ts.bucketed
    s = slice(rank, rank + 1)
    ts.set_batch(latent[s].cuda(), noise[s].cuda(), t[s].cuda(), mask[s].cuda(), time_ids=tid[s].cuda(), ids=[ids[s]] * 2, caption_token_lists=lists[rank:rank + 1])
    ts.capture(warmup=1)
    assert ts.graphs_frozen is None          # no frozen-TI variants under data parallelism (_run never takes them there)
    rows0 = ts.ti.params.clone()
    losses = []
    for i in range(4):
        ts.run(2e-4, lr_ti=1e-3)
        losses.append(float(ts.loss))
    ts.run(2e-4, lr_ti=0.0)                  # frozen token rows under DDP: the full graphs with lr 0 (no fast path), the exchange still runs
    gn = ts.grad_norms()
    torch.cuda.synchronize()
    return (rank, tr.params.cpu().numpy().copy(), losses, ts.ti.params.cpu().numpy().copy(), bool((ts.ti.params != rows0).any()), gn)


@pytest.mark.parametrize("zero1", [True, False])
def test_ddp_full_finetune_with_textual_inversion_captured(ddp_results, zero1):
    res = _scenario(ddp_results, "ti", zero1)
    assert torch.equal(torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])), "UNet replicas diverged"
    assert torch.equal(torch.from_numpy(res[0][3]), torch.from_numpy(res[1][3])), "token rows diverged"
    assert res[0][4] and res[1][4], "token rows did not train"
    for _, _, losses, _, _, gn in res:
        assert all(x == x for x in losses) and gn["unet"] > 0 and gn["unet"] == gn["unet"]
    assert abs(res[0][5]["unet"] - res[1][5]["unet"]) <= 1e-5 * res[0][5]["unet"]        # the read-out is rank-independent
