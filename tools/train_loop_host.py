"""Host-side time of the train() loop's per-step calls (set_batch, TrainStep.run -> graph replay): does the host run ahead of the GPU?
  python tools/train_loop_host.py [steps]"""
import os, sys, time, tempfile, shutil, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd.config import TrainingConfig
from sd_lora_trainer_amd.train import train
import sd_lora_trainer_amd.step as S

acc = collections.defaultdict(list)
def timed(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        acc[name].append(time.perf_counter() - t0)
        return r
    setattr(cls, name, f)
for n in ("set_batch", "set_hyper", "_run"):
    timed(S.TrainStep, n)
og = torch.cuda.CUDAGraph.replay
def rp(self):
    t0 = time.perf_counter(); og(self); acc["graph.replay"].append(time.perf_counter() - t0)
torch.cuda.CUDAGraph.replay = rp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tmp = tempfile.mkdtemp(prefix="sdlt_tl_")
cfg = TrainingConfig(lora_training_urls="synthetic:8", concept_mode="object", pretrained_model={"path": "synthetic:sdxl"}, seed=0, resolution=1024,
                     train_batch_size=1, max_train_steps=n, lora_rank=16, output_dir=tmp, n_sample_imgs=0, unet_lr=1e-3, ti_lr=1e-3)
gen = train(cfg, every_step=True)
t_iter = []
t0 = time.perf_counter()
try:
    while True:
        next(gen)
        t1 = time.perf_counter(); t_iter.append(t1 - t0); t0 = t1
except StopIteration as e:
    done, _ = e.value
shutil.rmtree(tmp, ignore_errors=True)
med = lambda v: sorted(v)[len(v) // 2] * 1e3
print(f"images/s {done.training_attributes['images_per_second']:.2f}; host ms per loop iteration (median) {med(t_iter[10:]):.3f}")
for k, v in acc.items():
    print(f"  {k:14s} median {med(v[10:]):8.3f} ms   max {max(v[10:]) * 1e3:8.3f} ms   n {len(v)}")
