"""Multi-GPU execution of the training path on one 8x MI355X node.

LoRA / textual-inversion jobs shard as INDEPENDENT units - one job per GPU, no exchange step:
the reference's own multi-job story is a shell script of `python main.py cfg_i.json` lines
(/root/reference scripts/create_hyperparam_sweep.py:135-152) whose processes race for "the GPU with most free
memory" (trainer/utils/utils.py:64-89).  Here job i is pinned to GPU i mod N through HIP_VISIBLE_DEVICES
*before* the process starts.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests) is used only
to bracket timed regions: a barrier and a MAX reduction of the elapsed time - never on the data path.
"""
import os
import subprocess
import sys

import torch


def job_env(job_index, n_gpus, base_env=None):
    """Environment for job `job_index` of a sweep on an `n_gpus` node: one visible device, chosen before start."""
    env = dict(base_env if base_env is not None else os.environ)
    gpu = job_index % n_gpus
    env["HIP_VISIBLE_DEVICES"] = str(gpu)
    env["CUDA_VISIBLE_DEVICES"] = str(gpu)          # PyTorch-ROCm honours either spelling
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def sweep_plan(config_paths, n_gpus, jobs_per_gpu=1):
    """[(config, gpu, wave)]: jobs run in waves of n_gpus * jobs_per_gpu; with jobs_per_gpu = 1 job i runs on GPU i mod n_gpus,
    otherwise consecutive groups of jobs_per_gpu configs share a GPU (and ONE process: train.train_concurrent)."""
    return [(c, (i // jobs_per_gpu) % n_gpus, i // (jobs_per_gpu * n_gpus)) for i, c in enumerate(config_paths)]


def run_sweep(config_paths, n_gpus, entry=("-m", "sd_lora_trainer_amd.train"), dry_run=False, jobs_per_gpu=1):
    """Launch a hyper-parameter sweep job-parallel; returns the list of (cmd, env) (and runs them unless dry_run).
    jobs_per_gpu > 1: each process gets that many configs and steps them concurrently on its GPU (one stream per job)."""
    groups = [config_paths[i:i + jobs_per_gpu] for i in range(0, len(config_paths), jobs_per_gpu)]
    launched = []
    for wave_start in range(0, len(groups), n_gpus):
        procs = []
        for i, grp in enumerate(groups[wave_start:wave_start + n_gpus], start=wave_start):
            cmd = [sys.executable, *entry, *grp]
            env = job_env(i, n_gpus)
            launched.append((cmd, {k: env[k] for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}))
            if not dry_run:
                procs.append(subprocess.Popen(cmd, env=env))
        for p in procs:
            p.wait()
    return launched


def init_distributed(backend=None):
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        torch.distributed.init_process_group(backend, **kw)
    return rank, world, local_rank


def barrier_sync():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (identity when not distributed)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t)


def aggregate_throughput(units_per_rank, elapsed_local, device="cpu"):
    """Whole-job throughput of N independent replicas: (sum of units) / (max elapsed)."""
    world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
    return world * units_per_rank / max_over_ranks(elapsed_local, device)
