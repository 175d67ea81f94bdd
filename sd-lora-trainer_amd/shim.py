"""Call-compatible seams - what makes the C-ABI kernels a DROP-IN behind the reference's own loop body rather than only
a parallel trainer.  The reference (/root/reference) talks to its model through duck-typed third-party interfaces; this module
presents the same ones over the engine:

  (1) model call          `unet(sample, timesteps, encoder_hidden_states=..., timestep_cond=None, added_cond_kwargs={"text_embeds",
                          "time_ids"}, return_dict=False)[0]`  (main.py:329-336), differentiable: `loss.backward()` (main.py:363)
                          reaches the adapters' `.grad` and the gradient w.r.t. the text conditioning through a
                          torch.autograd.Function whose backward is the engine's explicit backward plan;
                          `.parameters()`, `.requires_grad_()`, `.device`, `.dtype`, `.save_pretrained()` (main.py:109,146,375;
                          checkpoint.py:175,212)
  (2) processor seam      `DAAMScores`: per hooked attn2 layer an object with `.cross_attention_scores [B, N, 77]`, in the order
                          `find_attnprocessor2_0` walks them (ti_cross_attn_loss.py:88-112, 244-246).  With `keep_daam_maps` the maps are
                          OUTPUTS of the same autograd node as the prediction ("kept in graph", ti_cross_attn_loss.py:201-212): the
                          reference's `DAAMLoss` / `compute_token_attention_loss` (ti_cross_attn_loss.py:239-268, loss.py:10-80) run on
                          them unchanged and `loss.backward()` carries every layer's own dS back into that layer's Q and K
                          (S = Q K^T / sqrt(d): dQ += dS K / sqrt(d), dK += dS^T Q / sqrt(d), two GEMMs per hooked layer in the
                          backward plan).  The fused step (step.TrainStep) shares ONE dS per resolution instead - same values
  (3) adapter API         `LoraConfig` + `get_peft_model` + `get_peft_model_state_dict` with peft's key names
                          `base_model.model.<module path>.lora_A.weight / lora_B.weight` (optimizer.py:86-95, checkpoint.py:183-184)
  (4) optimizer API       any `torch.optim.Optimizer` over `unet.parameters()`: the parameters ARE the fp32 master copies of the
                          adapter arena (shared storage), the bf16 compute copies are refreshed at the next forward
                          (the fused device-side AdamW of `optimizer.OptimizerCollection` is the fast path, this is the compatible one)

The fused single-graph step (step.TrainStep) stays the fast path; this one pays torch's autograd bookkeeping and a few copies
per call, and exists so that reference-shaped code runs unchanged.
"""
import dataclasses
import json
import os
from typing import List

import torch

from . import topology
from .unet import CTX_PAD, F32, Runtime, UNet


@dataclasses.dataclass
class LoraConfig:
    """peft.LoraConfig's fields the reference sets (optimizer.py:86-95)."""
    r: int = 16
    lora_alpha: float = 16.0
    init_lora_weights: str = "gaussian"
    target_modules: List[str] = dataclasses.field(default_factory=lambda: ["to_k", "to_q", "to_v", "to_out.0", "conv2"])
    use_dora: bool = False


try:                                    # with diffusers installed the default processors ARE diffusers' class, so the reference's
    from diffusers.models.attention_processor import AttnProcessor2_0          # `isinstance(module, AttnProcessor2_0)` holds as it stands
except Exception:                       # (this image has no diffusers: a class of the same name)
    class AttnProcessor2_0:
        """Default processor object of an attention node (diffusers' `AttnProcessor2_0` in name and role): the engine's fused
        attention kernels do the work, the object only marks the place where a replacement can be installed."""


class DAAMScores(AttnProcessor2_0):
    """What sits at `<...>.attn2.processor` until something else is installed there.  Whatever object is found at that attribute after
    a forward - this one, or the reference's `DAAMLossAttnProcessor2_0(name)` put there by `init_daam_loss` with `setattr`
    (ti_cross_attn_loss.py:336-364) - receives `.cross_attention_scores` = sum_heads(Q K^T / sqrt(d)) [B, N, 77] of that layer: fp32 and,
    like the reference's (ti_cross_attn_loss.py:197-212), part of the autograd graph of the call that produced it."""

    def __init__(self, name=None):
        self.name, self.cross_attention_scores = name, None


class _Node:
    """One level of the module tree the reference walks by dotted names (`get_module_by_name` = reduce(getattr, name.split(".")),
    ti_cross_attn_loss.py:326-333): children by attribute (`node.attentions`), by string index (`getattr(node, "0")`, what the dotted walk
    does on an nn.ModuleList) and by `node[0]`; a missing child raises AttributeError like a module would."""

    def __init__(self):
        object.__setattr__(self, "_children", {})

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "_children")[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self._children[name] = value

    def __getitem__(self, i):
        return self._children[str(i)]

    def __len__(self):
        return len(self._children)

    def __iter__(self):
        return iter(self._children.values())

    def _descend(self, path):
        node = self
        for part in path:
            if part not in node._children:
                node._children[part] = _Node()
            node = node._children[part]
        return node


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, sample, timesteps, ehs, text_embeds, time_ids, *params):
        rt, u = mod.rt, mod.unet
        B, _, h, w = sample.shape
        assert B == rt.B, f"the module was built for batch {rt.B}"
        if mod._dirty:                  # an optimizer stepped the fp32 masters since the last forward
            u.arena.refresh_shadows()
            mod._dirty = False
        x64 = mod._buf("x64", B * h * w, 64)
        x64.zero_()
        x64[:, :4] = sample.detach().permute(0, 2, 3, 1).reshape(B * h * w, 4).to(x64.dtype)
        ctxb = mod._buf("ctx", B * CTX_PAD, u.cfg["cross_dim"])
        ctxb.zero_()
        ctxb.view(B, CTX_PAD, -1)[:, :77] = ehs.detach().to(ctxb.dtype)
        pooled = tid = None
        if u.cfg["addition"]:
            pooled = text_embeds.detach().to(rt.act)
            tid = time_ids.detach().reshape(-1).to(device=rt.device, dtype=F32)
        rt.want_dpooled = bool(u.cfg["addition"] and text_embeds is not None and text_embeds.requires_grad)
        rt.keep_daam_maps = mod.keep_daam_maps
        pred = u.forward(x64, timesteps.detach().to(device=rt.device, dtype=F32), ctxb, pooled, tid, B=B, H=h, W=w)
        maps = []
        if mod.keep_daam_maps:          # the score maps leave as further outputs of this node: a loss on them back-propagates (seam 2)
            maps = [S[:, :, :77].clone() for _, S in rt.daam]
            ctx.map_names = [name for name, _ in rt.daam]
        ctx.n_maps = len(maps)
        ctx.mod, ctx.shape = mod, (B, h, w)
        ctx.ehs_dtype, ctx.te_dtype = ehs.dtype, (text_embeds.dtype if text_embeds is not None else None)
        return (pred.view(B, h, w, 4).permute(0, 3, 1, 2).to(sample.dtype).contiguous(), *maps)

    @staticmethod
    def backward(ctx, dpred, *dmaps):
        mod = ctx.mod
        rt, u = mod.rt, mod.unet
        B, h, w = ctx.shape
        if dpred is None:               # (a loss on the maps alone)
            dpred = torch.zeros(B, 4, h, w, device=rt.device)
        # gradients of the score maps -> per-layer (dS, dS^T) operands of the backward plan's score-gradient GEMMs
        rt.daam_layer_grads = None
        if ctx.n_maps and any(g is not None for g in dmaps):
            lay = {}
            for name, g in zip(ctx.map_names, dmaps):
                if g is None:
                    continue
                N = g.shape[1]
                dS = mod._buf(("dS", name), B * N, CTX_PAD)
                dS.zero_()
                dS.view(B, N, CTX_PAD)[:, :, :77] = g.to(dS.dtype)
                dSt = mod._buf(("dSt", name), B * CTX_PAD, N)
                dSt.view(B, CTX_PAD, N).copy_(dS.view(B, N, CTX_PAD).transpose(1, 2))
                lay[name] = (dS, dSt)
            rt.daam_layer_grads = lay
        d64 = mod._buf("dpred64", B * h * w, 64)
        d64.zero_()
        d64[:, :4] = dpred.permute(0, 2, 3, 1).reshape(B * h * w, 4).to(d64.dtype)
        dctx = mod._buf("dctx", B * CTX_PAD, u.cfg["cross_dim"])
        dctx.zero_()
        rt.daam_grads, rt.daam_applied = None, False
        with torch.enable_grad():       # (irrelevant to the HIP ops; the CPU op emulation of the tests differentiates with autograd inside)
            u.backward(d64, dctx)
        rt.daam_layer_grads = None
        g_ehs = dctx.view(B, CTX_PAD, -1)[:, :77].to(ctx.ehs_dtype).clone()
        g_te = None
        if rt.want_dpooled:
            P = u.cfg["proj_class_in"] - 6 * u.cfg["addition_time_embed_dim"]
            g_te = u.dadd_in[:, :P].to(ctx.te_dtype).clone()
        grads = []
        for e in u.arena.entries:       # same order as UNetModule.parameters(): A then B (then the DoRA magnitude) of every adapted layer
            grads += [e["gA"].clone(), e["gB"].clone()] + ([e["gM"].clone()] if u.arena.dora else [])
        return (None, None, None, g_ehs, g_te, None, *grads)


class UNetModule:
    """UNet2DConditionModel-shaped object over the engine's UNet plan (see the module docstring)."""

    def __init__(self, version_or_cfg, state_dict, lora_config: LoraConfig = None, *, batch_size=1, device="cuda:0", runtime=None):
        cfg = topology.CONFIGS[version_or_cfg] if isinstance(version_or_cfg, str) else version_or_cfg
        self.rt = runtime or Runtime(device, batch_size)
        lc = lora_config or LoraConfig()
        assert sorted(lc.target_modules) == sorted(["to_k", "to_q", "to_v", "to_out.0", "conv2"]), "the fused kernels adapt the reference's target set"
        self.peft_config = lc
        self.unet = UNet(self.rt, cfg, state_dict, lora_rank=lc.r, lora_alpha_multiplier=lc.lora_alpha / lc.r, use_dora=lc.use_dora)
        self.config = dict(cfg)
        self.keep_daam_maps = False
        # seam 2 as the reference installs it: `down_blocks.i.attentions.j.transformer_blocks.k.attn2.processor` (and up_blocks / mid_block)
        # are real attributes; `find_attnprocessor2_0` / `init_daam_loss` (ti_cross_attn_loss.py:88-112, 336-364) walk and replace them
        self._tree = _Node()
        self._hooked_nodes = []
        for a in self.unet.cross_attns:
            node = self._tree._descend(a.name.split("."))
            node.processor = DAAMScores(a.name + ".processor")
            if a.hooked:                # (the mid block's attn2 exists in the tree but the reference never walks it: no score map is produced there)
                self._hooked_nodes.append(node)
        self._params, self._names = [], []
        for e in self.unet.arena.entries:
            for key, nm in (("A", "lora_A.weight"), ("B", "lora_B.weight")) + ((("M", "lora_magnitude_vector"),) if lc.use_dora else ()):
                p = torch.nn.Parameter(e[key], requires_grad=True)      # shares storage with the arena's fp32 master
                self._params.append(p)
                self._names.append(f"base_model.model.{e['name']}.{nm}")
        self._dirty, self._bufs = True, {}

    # ---- nn.Module-shaped surface ------------------------------------------------------------------------------------
    def __getattr__(self, name):        # (only reached for names that are not ordinary attributes: the block lists of the module tree)
        tree = self.__dict__.get("_tree")
        if tree is not None and name in tree._children:
            return tree._children[name]
        raise AttributeError(name)

    @property
    def daam_processors(self):
        """The objects currently installed at the hooked attn2 layers' `.processor`, in the order `find_attnprocessor2_0` finds them."""
        return [n.processor for n in self._hooked_nodes]

    device = property(lambda self: self.rt.device)
    dtype = property(lambda self: self.rt.act)

    def parameters(self):
        return iter(self._params)

    def named_parameters(self):
        return iter(zip(self._names, self._params))

    def requires_grad_(self, flag=True):
        for p in self._params:
            p.requires_grad_(flag)
        return self

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def init_lora_weights(self, generator=None):
        """peft init_lora_weights="gaussian": A ~ N(0, (1/r)^2), B = 0."""
        r = self.peft_config.r
        with torch.no_grad():
            for e in self.unet.arena.entries:
                e["A"].copy_(torch.randn(e["A"].shape, generator=generator, device=e["A"].device) / r)
                e["B"].zero_()
        self._dirty = True

    def _buf(self, key, *shape):
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self.rt.zeros(*shape)
            self._bufs[key] = t
        return t

    def __call__(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, added_cond_kwargs=None, return_dict=False, **kw):
        assert timestep_cond is None, "timestep_cond is None in the reference's call (main.py:333)"
        add = added_cond_kwargs or {}
        te, tid = add.get("text_embeds"), add.get("time_ids")
        if self.unet.cfg["addition"]:
            assert te is not None and tid is not None, "SDXL needs added_cond_kwargs = {'text_embeds', 'time_ids'}"
        else:
            te = tid = None
        vers = [p._version for p in self._params]          # an optimizer step (in-place update of a parameter) bumps its version
        if vers != getattr(self, "_versions", None):
            self._dirty = True
        self._versions = vers
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep] * sample.shape[0])
        procs = self.daam_processors
        installed = any(type(p) is not DAAMScores for p in procs)      # something was put there with setattr (init_daam_loss): it wants its maps
        keep, self.keep_daam_maps = self.keep_daam_maps, (self.keep_daam_maps or installed)
        try:
            out, *maps = _UNetFn.apply(self, sample, timestep, encoder_hidden_states, te, tid, *self._params)
        finally:
            self.keep_daam_maps = keep
        for proc, S in zip(procs, maps):
            proc.cross_attention_scores = S
        if return_dict:
            import types
            return types.SimpleNamespace(sample=out)
        return (out,)

    # ---- adapter API (peft) ------------------------------------------------------------------------------------------------
    def get_peft_model_state_dict(self):
        """peft.get_peft_model_state_dict: {base_model.model.<path>.lora_A.weight: [r, Cin(,3,3)], ...lora_B.weight: [Cout, r(,1,1)]}."""
        out = {}
        for name, (A, B, *m) in self.unet.arena.export().items():
            out[f"base_model.model.{name}.lora_A.weight"] = A
            out[f"base_model.model.{name}.lora_B.weight"] = B
            if m:
                out[f"base_model.model.{name}.lora_magnitude_vector"] = m[0]
        return out

    def save_pretrained(self, output_dir):
        """peft `save_pretrained` (checkpoint.py:175): adapter_config.json + adapter_model.safetensors."""
        from safetensors.torch import save_file
        os.makedirs(output_dir, exist_ok=True)
        lc = self.peft_config
        with open(os.path.join(output_dir, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LORA", "r": lc.r, "lora_alpha": lc.lora_alpha, "init_lora_weights": lc.init_lora_weights,
                       "target_modules": list(lc.target_modules), "use_dora": lc.use_dora}, f, indent=2)
        save_file({k: v.contiguous() for k, v in self.get_peft_model_state_dict().items()}, os.path.join(output_dir, "adapter_model.safetensors"))


def get_peft_model(version_or_cfg, state_dict, lora_config, **kw):
    """optimizer.py:86-95 `get_peft_model(unet, LoraConfig(...))` for this engine: the adapted model is built from the frozen weights."""
    return UNetModule(version_or_cfg, state_dict, lora_config, **kw)


def get_peft_model_state_dict(model):
    return model.get_peft_model_state_dict()
