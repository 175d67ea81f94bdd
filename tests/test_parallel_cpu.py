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
