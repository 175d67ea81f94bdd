"""Is the LoRA + TI step bit-reproducible on the REAL SDXL topology at the headline size?  Runs forward_backward() twice from the same state and compares every persistent
buffer of every plan module (activations, gradients) bit for bit, listing the buffers that differ in creation order (a module's buffers are created the first time its
forward / backward runs): the first differing one names the kernel that is not deterministic.  usage: determinism_probe.py [sdxl|sd15] [h] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import unet_ref as U
from tests import test_real_topology_gpu as T
from sd_lora_trainer_amd import unet as unet_mod

ORDER = []          # (module, key) in the order the persistent buffers come into existence = execution order of the first pass
_buf = unet_mod._Module.buf


def _buf_logged(self, key, *shape, **kw):
    if key not in self._b:
        ORDER.append((self, key))
    return _buf(self, key, *shape, **kw)


unet_mod._Module.buf = _buf_logged
version = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
h = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = U.CONFIGS[version]
xl = cfg["addition"]
sd = T._unet_state(version)
lora = U.init_lora(cfg, 16, seed=1, b_std=0.02)
hf = [T._hf_clip("clip_l", 11), T._hf_clip("clip_g", 12)] if xl else [T._hf_clip("clip_l", 11)]
rt, unet, ts = T._build_product(version, B, h, sd, lora, hf, 16, token_attention_loss_w=2e-2)
b = T._batch(cfg, B, h, 3, [10, 900, 500, 999])
T._set(ts, b, xl, h, len(hf))


def modules(root, seen, out, prefix):
    if id(root) in seen:
        return
    seen.add(id(root))
    if isinstance(root, unet_mod._Module):
        out.append((prefix, root))
    for k, v in list(vars(root).items()):
        if isinstance(v, unet_mod._Module):
            modules(v, seen, out, f"{prefix}.{k}")
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                if isinstance(x, unet_mod._Module) or hasattr(x, "__dict__") and any(isinstance(y, unet_mod._Module) for y in vars(x).values()):
                    modules(x, seen, out, f"{prefix}.{k}[{i}]")
        elif hasattr(v, "__dict__") and not isinstance(v, (torch.Tensor, type)) and k in ("text", "ta", "ti", "encoders"):
            modules(v, seen, out, f"{prefix}.{k}")


def snapshot():
    snap = {}
    for i, (m, k) in enumerate(ORDER):
        t = m._b.get(k)
        if isinstance(t, torch.Tensor):
            snap[f"{i:05d} {m.name}:{k}"] = t.clone()
    snap["arena.grads"] = unet.arena.grads.clone()
    for i, r in enumerate(ts.ti.grad_rows):
        snap[f"ti.grad_rows[{i}]"] = r.clone()
    snap["ctx"] = ts.ctx.clone()
    snap["dctx"] = ts.dctx.clone()
    return snap


for tag in ("eager", "eager"):
    ts.forward_backward()
torch.cuda.synchronize()
a = snapshot()
ts.forward_backward()
torch.cuda.synchronize()
bsn = snapshot()
diff = [(k, float((a[k].float() - bsn[k].float()).abs().max()), float(a[k].float().abs().max())) for k in a if k in bsn and a[k].shape == bsn[k].shape and not torch.equal(a[k], bsn[k])]
print(f"{len(a)} buffers compared, {len(diff)} differ between two eager passes from the same state")
first = min(int(k.split()[0]) for k, _, _ in diff if k[0].isdigit()) if diff else -1
print("first differing buffer in execution order:", first)
for k in sorted(a):
    if k[0].isdigit() and first - 25 <= int(k.split()[0]) <= first + 40:
        same = k in bsn and a[k].shape == bsn[k].shape and torch.equal(a[k], bsn[k])
        print(f"  {k:90s} {'same' if same else 'DIFFERS'}  {tuple(a[k].shape)} {a[k].dtype}")
for k, d, m in diff[:400]:
    print(f"  {k:90s} max abs diff {d:.3e} (max abs {m:.3e})")
# NaN-aware recount: torch.equal is False for NaN == NaN; list the buffers whose only 'difference' is NaN / uninitialised padding
nan_only = [k for k, d, m in diff if d != d]
print("buffers with NaN in the comparison (uninitialised padding?):", len(nan_only), nan_only[:10])
