"""One optimisation step of the reference's training loop (/root/reference main.py:263-382) as a static plan
over the HIP kernels: add-noise -> UNet(+LoRA) forward -> masked/SNR-weighted MSE -> explicit backward ->
grouped LoRA gradients -> fused AdamW(+L1) -> bf16 shadow refresh.  On the GPU the whole body is captured
once into a hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm) and replayed; per-step scalars (learning
rates, bias corrections) live in a small device buffer that is updated before each replay.
"""
import math
import os

import torch

from .clip import T_TOKENS, TP
from .daam import TokenAttentionLoss
from .ti import TiState
from .unet import CTX_PAD, F32, Runtime, UNet

assert TP == CTX_PAD


TEXT_PAIR = os.environ.get("SDLT_TEXT_PAIR", "1") != "0"




class TextStack:
    """The text encoders wired as diffusers' `encode_prompt` wires them (trainer/inference.py:131-177):
    SD1.5: prompt_embeds = CLIP-L last_hidden_state (after final LN).
    SDXL : prompt_embeds = concat(CLIP-L hidden_states[-2], bigG hidden_states[-2]); pooled = bigG text_embeds."""

    def __init__(self, rt, encoders, pool_mode="argmax", eos_token_id=49407, concurrent=False, arena=None):
        """arena: the (finalized) LoraArena the encoders' q/k/v/out_proj adapters live in, when the text encoders are
        LoRA-trained too (`text_encoder_lora_optimizer`, trainer/optimizer.py:157-202); None otherwise."""
        self.rt, self.encoders, self.pool_mode, self.eos = rt, encoders, pool_mode, eos_token_id
        self.arena, self._grad_plan = arena, None
        B = rt.B
        self.ids = [torch.zeros(B, T_TOKENS, dtype=torch.int64, device=rt.device) for _ in encoders]
        self.pool_rows = torch.zeros(B, dtype=torch.int64, device=rt.device)
        self.widths = [e.D for e in encoders]
        # concurrent=True runs the encoders on forked streams (and the step then replays three small hipGraphs, because
        # ROCm's executor only overlaps branches forked near the root of a small graph).  Measured on the SDXL step it LOSES
        # 1.7 ms: each fork/join costs ~0.7 ms of queue hand-off plus ~90 us at every cross-queue edge, more than CLIP-L
        # (2.6 ms, mostly hidden) saves - so the default is one stream, one graph.
        self.concurrent = bool(concurrent) and torch.device(rt.device).type == "cuda"
        self.side = [torch.cuda.Stream(device=rt.device) for _ in encoders] if self.concurrent else []
        # two fused encoders (SDXL): layer i of CLIP-L rides in the launches of layer i of bigG (ops.run_paired; SDLT_TEXT_PAIR=0: one after the other)
        self.paired = (TEXT_PAIR and len(encoders) == 2 and all(getattr(e, "fused", False) for e in encoders) and not self.concurrent
                       and torch.device(rt.device).type == "cuda")

    def pool_position_table(self, ids_table):
        """Pooling position of every row of an id table [n, 77] (device) - train() gathers the batch's positions from it per step."""
        return ids_table.argmax(-1) if self.pool_mode == "argmax" else (ids_table == self.eos).int().argmax(-1)

    def set_ids_from_tables(self, tables, sel, pool_pos_table):
        """The step's ids as gathers from per-caption tables on the device (train()): one launch per encoder + two for the pooling rows
        (set_ids: a copy per encoder and five launches for argmax / arange / scale / add / copy)."""
        for dst, tab in zip(self.ids, tables):
            if tab.dtype == dst.dtype and tab.shape[1:] == dst.shape[1:]:
                torch.index_select(tab, 0, sel, out=dst)
            else:                                        # (an id table from a cache that holds int32 ids: index_select's out= does not convert)
                dst.copy_(tab[sel])
        if getattr(self, "_pool_base", None) is None:
            self._pool_base = torch.arange(self.pool_rows.shape[0], device=self.pool_rows.device) * TP
            self._pool_tmp = torch.zeros_like(self.pool_rows)
        if pool_pos_table.dtype == self._pool_tmp.dtype:
            torch.index_select(pool_pos_table, 0, sel, out=self._pool_tmp)
        else:
            self._pool_tmp.copy_(pool_pos_table[sel])
        torch.add(self._pool_base, self._pool_tmp, out=self.pool_rows)

    def set_ids(self, ids_per_encoder):
        for dst, src in zip(self.ids, ids_per_encoder):
            dst.copy_(src)
        last = self.ids[-1]              # on the device: no host round trip in the step loop
        # transformers CLIPTextModel pooling: legacy configs (eos_token_id == 2, SDXL's text_encoder_2) take argmax(input_ids),
        # newer ones the first position equal to eos_token_id
        pos = last.argmax(-1) if self.pool_mode == "argmax" else (last == self.eos).int().argmax(-1)
        self.pool_rows.copy_(torch.arange(last.shape[0], device=last.device) * TP + pos)

    def _fan_out(self, jobs):
        """Run one job per encoder, each on its own side stream, joined before returning.  The encoders are
        independent and each of their kernels fills a fraction of the chip (M = 128 rows), so CLIP-L rides along bigG
        for free.  Works eagerly and inside hipGraph capture (fork/join become graph edges)."""
        if not self.side:
            for job in jobs:
                job()
            return
        cur = torch.cuda.current_stream(self.rt.device)
        for s in self.side:
            s.wait_stream(cur)
        # every branch gets its own stream and the calling stream only waits: ROCm's graph executor releases a forked
        # branch when the forking queue reaches its *next* marker, so a branch left on the calling stream would hold the
        # other one back until it is done (observed in the kernel trace)
        for s, job in zip(self.side, jobs):
            with torch.cuda.stream(s):
                job()
        for s in self.side:
            cur.wait_stream(s)

    def forward(self, ctx):
        out = [None] * len(self.encoders)
        if self.paired:
            gens, off = [], 0
            for e, ids, w in zip(self.encoders, self.ids, self.widths):
                gens.append(e.forward_steps(ids, self.rt.B, hidden_out=ctx[:, off:off + w], pool_rows=self.pool_rows))
                off += w
            vals = self.rt.ops.run_paired(gens, min)          # both start at layer 0: in lockstep until the shorter encoder is done
            pooled = None
            for _, p in vals:
                pooled = p if p is not None else pooled
            return pooled
        jobs, off = [], 0
        for i, (e, ids, w) in enumerate(zip(self.encoders, self.ids, self.widths)):
            def job(i=i, e=e, ids=ids, off=off, w=w):
                _, out[i] = e.forward(ids, self.rt.B, hidden_out=ctx[:, off:off + w], pool_rows=self.pool_rows)
            jobs.append(job)
            off += w
        self._fan_out(jobs)
        pooled = None
        for p in out:
            pooled = p if p is not None else pooled
        return pooled

    def backward(self, dctx, d_pooled, grad_rows):
        jobs, off = [], 0
        if self.paired:
            gens = []
            for e, w, g in zip(self.encoders, self.widths, grad_rows):
                gens.append(e.backward_steps(dctx[:, off:off + w], d_pooled if e.with_projection else None, g))
                off += w
            self.rt.ops.run_paired(gens, max)                 # the deeper encoder runs alone down to the other's top layer, then in lockstep
        else:
            for e, w, g in zip(self.encoders, self.widths, grad_rows):
                def job(e=e, off=off, w=w, g=g):
                    e.backward(dctx[:, off:off + w], d_pooled if e.with_projection else None, g)
                jobs.append(job)
                off += w
            self._fan_out(jobs)
        if self.arena is not None:                     # dA / dB of every text-encoder adapter in one grouped launch
            if self._grad_plan is None:
                self._grad_plan = self.rt.ops.LoraGradPlan(self.arena.problems, self.arena.Rp, self.rt.device)
            self._grad_plan.run()
            if not getattr(self, "defer_dora_mag_grad", False):      # (TrainStep._tok_cond_reg runs it behind its second pass)
                self.arena.dora_mag_grad()


def ddpm_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """diffusers DDPMScheduler(beta_schedule="scaled_linear") table used by add_noise / compute_snr."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class ProdigyState:
    """Device-resident state of ONE Prodigy parameter group over a flat fp32 arena (sdlt_prodigy_step): p0, s and the
    scalars d, d_max, d_numerator, k.  Constructor values are the ones the reference passes (trainer/optimizer.py:24-34,
    135-145): betas (0.9, 0.99), decoupled decay, bias correction and safeguard warm-up on, d0 = 1e-6, eps = 1e-8."""

    def __init__(self, rt, params, *, d_coef=1.0, growth_rate=float("inf"), weight_decay=0.0, betas=(0.9, 0.99), eps=1e-8, d0=1e-6):
        self.rt, self.params = rt, params
        self.p0, self.s = torch.zeros_like(params), torch.zeros_like(params)
        self.state = rt.zeros(16, dtype=F32)
        self.acc = torch.zeros(2, dtype=torch.float64, device=params.device)
        self.d0, self.betas = d0, betas
        self.const = [betas[0], betas[1], math.sqrt(betas[1]), eps, weight_decay, d_coef, growth_rate]
        self.reset()

    def reset(self):
        """Back to step 0; p0 is taken from the parameters when the first step is queued."""
        self.s.zero_()
        self.state.zero_()
        self.state[:3] = self.d0
        self.started = False

    def before_run(self):
        if not self.started:
            self.p0.copy_(self.params)
            self.started = True

    def hyper_row(self, lr, l1_coef):
        return [lr, *self.const, l1_coef, 1.0, 1.0, 1.0, 1.0]

    def step(self, grads, m, v, hyper, l1_sum):
        self.rt.ops.prodigy_step(self.params, grads, self.p0, m, v, self.s, hyper, self.state, self.acc, l1_sum)

    def group(self, lr):
        """The keys `get_current_lr` reads from a Prodigy param group (trainer/optimizer.py:206-234); one device sync."""
        st = self.state.tolist()
        return dict(d=st[0], d0=st[1], d_max=st[2], d_numerator=st[3], d_denom=st[4], d_hat=st[5], k=int(st[6]), lr=lr,
                    betas=self.betas, use_bias_correction=True)


class TrainStep:
    def __init__(self, rt: Runtime, unet: UNet, *, latent_hw, snr_gamma=5.0, v_prediction=False, l1_penalty=0.03,
                 weight_decay=0.004, grad_accum=1, betas=(0.9, 0.999), eps=1e-8, text: TextStack = None, n_tokens=3,
                 token_attention_loss_w=3e-7, ti_weight_decay=0.0, ti_std_loss_w=0.01, optimizer="adamw", ti_optimizer="adamw",
                 prodigy_d_coef=1.0, prodigy_growth_rate=1.05, text_lora_weight_decay=1e-5, process_group=None,
                 cond_reg_w=0.0, tok_cov_reg_w=0.0, cond_target_norm=None, tok_cond_reg_w=0.0, reg_caption_ids=None, ddp_wire_dtype=None, ddp_zero1=None, ddp_force=False, ti_trainable=True):
        """ti_trainable=False (disable_ti with text-encoder LoRA, main.py:116-133): the text encoders run inside the step for their adapters' gradients, the token
        rows never move (their learning rate is forced to 0) and the logged total leaves the token regularisers out.
        process_group: data-parallel full fine-tune only (`unet.trainer` set) - a torch.distributed group (or True for the
        default one) over which the gradient arena is all-reduced once per optimiser step (RCCL on the GPU)."""
        adam8 = optimizer == "AdamW8bit"
        if adam8:
            # bitsandbytes' AdamW8bit (optimizer.py:19-21, the full fine-tune example) is AdamW with block-quantised moments.  Full fine-tune on one GPU (or
            # data parallel with the all-reduce exchange): the matrices' moments are held that way (fullft.WeightTrainer.enable_8bit - 12 fewer bytes per
            # parameter and step); with the sharded optimizer (ZeRO-1) the owned slices' moments, in blocks of 2048 consecutive elements (enable_zero1(adam8=True)).
            # LoRA / TI groups (a few MB) keep fp32 moments.
            optimizer = "adamw"
        if optimizer not in ("adamw", "prodigy"):
            raise NotImplementedError(f"Invalid optimizer_name for unet: {optimizer}")
        if ti_optimizer not in ("adamw", "prodigy"):
            raise NotImplementedError(f"Invalid optimizer_name: '{ti_optimizer}'")
        self.rt, self.unet = rt, unet
        self.ti_trainable = bool(ti_trainable)
        # the trained parameter group of the UNet: the LoRA arena, or every weight (full fine-tune, main.py:144-149)
        self.full_ft = getattr(unet, "trainer", None) is not None
        self.group = unet.trainer if self.full_ft else unet.arena
        if self.full_ft:
            l1_penalty = 0.0       # main.py:353: the L1 term only exists over unet_lora_parameters
        self.pg, self.world = None, 1
        # self.ddp: the data-parallel code path is live.  ddp_force (or SDLT_DDP_FORCE=1): also with ONE rank - the exact call sequence of the exchange step (per-bucket
        # graphs, in-place reduce-scatter / all-gather on slices of the arenas, asynchronous works joined around the graphs) on a 1-rank group, so that the first
        # multi-GPU run of `bench.py --full-ft --gpus N` is a measurement, not the first execution of the path (bench.py --dry-collectives).
        self.ddp = False
        self.coll_log = None       # a list: every collective of a step is recorded as (op, input offset, input numel, output offset, output numel, dtype)
        if process_group is not None:
            import torch.distributed as dist
            assert self.full_ft, "LoRA / TI jobs are independent per GPU (no collective); only the full fine-tune is data parallel"
            self.pg = None if process_group is True else process_group
            self.world = dist.get_world_size(self.pg)
            self.ddp = self.world > 1 or bool(ddp_force) or os.environ.get("SDLT_DDP_FORCE", "0") == "1"
            # bucketed, overlapped gradient exchange: the weight gradients are produced bucket by bucket at the end of the
            # backward (fullft.WeightTrainer.flush(bucket=i)), each bucket's all-reduce starts as soon as its gradients exist
            self.bucketed = self.ddp and grad_accum == 1
            unet.trainer.defer_flush = self.bucketed
            # ddp_wire_dtype "bf16" (or SDLT_DDP_WIRE=bf16): the matrix gradients cross xGMI as bf16 - half the wire bytes (per GPU 2 (N-1)/N x 5.1
            # instead of 10.3 GB for SDXL) for two more HBM passes per bucket (pack after its weight-gradient GEMMs, unpack after its
            # all-reduce: ~15 GB of traffic each way per step) and a bf16 sum over the ranks; the vector region (biases, norm affine)
            # and the token rows stay fp32.  Default fp32: exact, and no multi-GPU box was available to measure which side wins.
            wire = ddp_wire_dtype or os.environ.get("SDLT_DDP_WIRE", "fp32")
            assert wire in ("fp32", "bf16"), wire
            self.wire = torch.empty(unet.trainer.n_mat, dtype=torch.bfloat16, device=rt.device) if (wire == "bf16" and self.bucketed) else None
            # the exchange step: reduce-scatter -> AdamW on this rank's 1 / world of every bucket -> all-gather of the masters -> operand refresh
            # (fullft.WeightTrainer.enable_zero1); ddp_zero1=False / SDLT_DDP_ZERO1=0: all-reduce + the full AdamW on every rank (A/B)
            z = ddp_zero1 if ddp_zero1 is not None else (os.environ.get("SDLT_DDP_ZERO1", "1") != "0")
            self.zero1 = bool(z) and self.bucketed and optimizer == "adamw"
            if self.zero1:
                unet.trainer.enable_zero1(dist.get_rank(self.pg), self.world, adam8=adam8 and os.environ.get("SDLT_ADAM8", "1") != "0")
        self.adam8 = adam8 and self.full_ft and os.environ.get("SDLT_ADAM8", "1") != "0"
        if self.adam8 and not getattr(self, "zero1", False):
            unet.trainer.enable_8bit()
        self.text, self.ta_w, self.ti_wd = text, token_attention_loss_w, ti_weight_decay
        self.ti = TiState(rt, text.encoders, n_tokens, ti_std_loss_w) if text is not None else None
        self.ta = TokenAttentionLoss(rt, n_tokens) if text is not None else None
        rt.want_dpooled = text is not None and unet.cfg["addition"]
        B, (h, w) = rt.B, latent_hw
        cfg = unet.cfg
        self.B, self.h, self.w = B, h, w
        self.snr_gamma, self.v_pred, self.l1_penalty, self.wd, self.grad_accum = snr_gamma, v_prediction, l1_penalty, weight_decay, grad_accum
        self.betas, self.eps = betas, eps
        dev = rt.device
        z = lambda *s, dtype=F32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self.latent, self.noise, self.mask = z(B, 4, h, w), z(B, 4, h, w), z(B, 4, h, w)
        self.noisy = z(B, 4, h, w)
        self.timesteps = z(B, dtype=torch.int64)
        self.timesteps_f = z(B)
        self.ctx = rt.zeros(B * CTX_PAD, cfg["cross_dim"])
        self.dctx = rt.zeros(B * CTX_PAD, cfg["cross_dim"])
        self.pooled = self.time_ids = None
        if cfg["addition"]:
            self.pooled = rt.zeros(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])
            self.time_ids = z(B * 6)
        self.acp = ddpm_alphas_cumprod().to(dev)
        self.x64 = rt.zeros(B * h * w, 64)
        self.dpred64 = rt.zeros(B * h * w, 64)
        self.sums, self.loss, self.l1_sum = z(B * 2 * 65), z(1), z(1)      # sums: [B,2] finals + up to 64 partial slices per sample
        self.hyper = z(16)
        self.opt_step = 0
        self.graph, self.graphs, self.graphs_frozen, self.graphs_frozen_cached = None, [], None, None
        self._cond_cached = False
        # optional Prodigy groups (trainer/optimizer.py:24-34: growth_rate = unet_prodigy_growth_factor, d_coef = prodigy_d_coef;
        # :135-145 for the token rows: d_coef 1, unbounded growth)
        self.prodigy = ProdigyState(rt, self.group.params, d_coef=prodigy_d_coef, growth_rate=prodigy_growth_rate,
                                    weight_decay=weight_decay) if optimizer == "prodigy" else None
        self.prodigy_ti = ProdigyState(rt, self.ti.params, weight_decay=ti_weight_decay) \
            if (self.ti is not None and ti_optimizer == "prodigy") else None
        # a21: text-encoder LoRA (AdamW only, optimizer.py:190-199): its own arena, learning rate and weight decay
        self.te_arena = text.arena if text is not None else None
        self.te_wd = text_lora_weight_decay
        self.te_hyper = z(16) if self.te_arena is not None else None
        # optional regularisers of ConditioningRegularizer (loss.py:196-251), both weighted 0 by default (config.py:75-77)
        self.cond_reg_w, self.tok_cov_reg_w = cond_reg_w, tok_cov_reg_w
        self.cond_target_norm = cond_target_norm if cond_target_norm is not None else (34.5 if unet.cfg["addition"] else 27.8)   # loss.py:182
        self.cond_reg_loss, self.cond_norm = z(1), z(1)
        # tok_cond_reg_w (loss.py:207-211, 241-251): the same norm target on the conditioning of FOUR fixed captions around the
        # trigger ("a photo of TOK", "TOK", ...), encoded with autograd every step -> a second pass of the text encoders at batch
        # 4 (plan clones sharing the weights and the token tables), its row gradients added to the step's
        self.tok_cond_reg_w = tok_cond_reg_w if text is not None else 0.0
        self.tok_reg_loss, self.tok_reg_norm = z(1), z(1)
        if self.tok_cond_reg_w > 0.0:
            from .unet import ArenaView, clone_plan
            assert reg_caption_ids is not None, "tok_cond_reg_w > 0 needs the token ids of the regularisation captions"
            # DoRA text-encoder adapters: the first pass's magnitude-gradient launch also scales the dB rows by m / norm, so it has to see the SUM of both passes'
            # dB - it is deferred behind the second pass's accumulating dA / dB launch, and the second pass adds its own magnitude gradients in one more launch
            # (sdlt_dora_grad_desc.accumulate); optimizer.py:157-202 + loss.py:207-211
            self._reg_dora_plan = None
            text.defer_dora_mag_grad = text.arena is not None and text.arena.dora
            nreg = reg_caption_ids[0].shape[0]
            self.reg_rt = Runtime(dev, nreg, act_dtype=rt.act, ops=rt.ops)
            # text-encoder LoRA: the second pass differentiates through the adapters too - its dA / dB problems live in a view of the arena
            # and are ADDED to the step's by a second grouped launch
            self.reg_arena, self._reg_grad_plan = (ArenaView(text.arena) if text.arena is not None else None), None
            self.reg_encoders = [clone_plan(e, self.reg_rt, arena=self.reg_arena) for e in text.encoders]
            self.reg_ids = [i.to(dev, torch.int64).contiguous() for i in reg_caption_ids]
            self.reg_ctx = rt.zeros(nreg * CTX_PAD, cfg["cross_dim"])
            self.reg_dctx = rt.zeros(nreg * CTX_PAD, cfg["cross_dim"])
        # gradient accumulation (main.py:362-366): every micro-step back-propagates loss / k; the optimizers step on the k-th
        # (or on the last batch of an epoch).  The kernels overwrite their gradient buffers, so micro-steps add them into
        # accumulators that are handed back at the boundary.
        self._micro, self._acc, self.graphs_micro = 0, None, None
        if grad_accum > 1:
            gs = [self.group.grads] + ([self.ti.grads] if self.ti is not None else []) + ([self.te_arena.grads] if self.te_arena is not None else [])
            self._acc = [(g_, torch.zeros_like(g_)) for g_ in gs]

    # -------------------------------------------------------------------------------- inputs
    def set_batch(self, latent, noise, timesteps, mask, ctx=None, pooled=None, time_ids=None, ids=None, caption_token_lists=None, caption_table=None):
        """latent/noise/mask [B,4,h,w] fp32, timesteps int64 [B]; SDXL: time_ids [B,6].
        Without text encoders: ctx [B,77,D] (+ pooled [B,P]) are the injected conditioning.
        With text encoders (textual inversion): ids = [input_ids [B,77] per tokenizer] and caption_token_lists[b] =
        tokenizer.encode(caption_b) (unpadded, for the token-attention loss, loss.py:32)."""
        # (train() fills the step's own buffers in place - `x is self.x` - and those copies fall away)
        for dst, src in ((self.latent, latent), (self.noise, noise), (self.mask, mask), (self.timesteps, timesteps)):
            if src is not dst:
                dst.copy_(src)
        self.timesteps_f.copy_(timesteps)              # (the copy converts: no float temporary)
        if self.text is None:
            self.ctx.view(self.B, CTX_PAD, -1)[:, :77].copy_(ctx)
            if self.pooled is not None:
                self.pooled.copy_(pooled)
        else:
            # cached conditioning (f4): once the token rows are frozen (ti lr == 0) and the text encoders carry no adapters, the
            # conditioning of a caption is a constant of the job - the caller may hand it over instead of having it re-encoded
            self._cond_cached = ctx is not None
            if self._cond_cached:
                self.ctx.view(self.B, CTX_PAD, -1)[:, :77].copy_(ctx)
                if self.pooled is not None:
                    self.pooled.copy_(pooled)
            elif isinstance(ids, tuple):               # (tables, rows, pooling-position table): gathers on the device
                self.text.set_ids_from_tables(*ids)
            else:
                self.text.set_ids(ids)
            if caption_table is not None:          # (table, rows): per-caption constants already on the device (train(): no host work per step)
                self.ta.set_from_table(*caption_table)
            else:
                if getattr(self, "_train_ids_host", None) is None:
                    self._train_ids_host = self.text.encoders[0].train_ids.tolist()
                self.ta.set_captions(caption_token_lists, self._train_ids_host)
        if self.time_ids is not None and time_ids is not None:
            # a job passes the same tensor every step: copied once.  The key is (object, storage, version counter), so a caller that
            # rewrites its tensor in place between steps (per-batch crop / size conditioning) is seen
            key = (id(time_ids), time_ids.data_ptr(), time_ids._version)
            if key != getattr(self, "_time_ids_key", None):
                self.time_ids.copy_(time_ids.reshape(-1))
                self._time_ids_key, self._time_ids_src = key, time_ids

    def set_hyper(self, lr, lr_ti=0.0, lr_te=0.0):
        """Host scalars of this optimiser step -> device buffers (see sdlt_adamw_fused).
        The upload is an async copy from a ring of pinned staging rows: a pageable-memory copy would block the host until the
        previous step has finished on the GPU, so the next graph launch could never be queued behind the running one (the
        GPU then idled ~1 ms per step while the host caught up)."""
        self.opt_step += 1
        b1, b2 = self.betas
        n = self.group.n
        bc = [1.0 - b1 ** self.opt_step, 1.0 - b2 ** self.opt_step]
        # data parallel: the all-reduce SUMS the ranks' gradients, the mean is the optimizer's gradient scale
        rows = [[lr, b1, b2, self.eps, self.wd, *bc, self.l1_penalty / n, 1.0 / self.world] if self.prodigy is None
                else self.prodigy.hyper_row(lr, self.l1_penalty / n)]
        if self.ti is not None:
            # data parallel: the ranks see different captions, so the token-row gradients are exchanged too (summed; mean via the scale)
            rows.append([lr_ti, b1, b2, self.eps, self.ti_wd, *bc, 0.0, 1.0 / self.world] if self.prodigy_ti is None
                        else self.prodigy_ti.hyper_row(lr_ti, 0.0))
        dsts = [self.hyper] + ([self.ti.hyper] if self.ti is not None else [])
        if self.te_arena is not None:
            rows.append([lr_te, b1, b2, self.eps, self.te_wd, *bc, 0.0, 1.0 / self.world])
            dsts.append(self.te_hyper)
        for pr in (self.prodigy, self.prodigy_ti):
            if pr is not None:
                pr.before_run()
        cuda = self.hyper.is_cuda
        if cuda and getattr(self, "_hyper_ring", None) is None:
            self._hyper_ring = torch.zeros(64, 3, 16, dtype=torch.float32).pin_memory()
            self._hyper_done = [None] * 64          # event after the copies of a slot: waited for before the slot is reused
            self._hyper_slot = 0
        if cuda and self._hyper_done[self._hyper_slot] is not None:
            self._hyper_done[self._hyper_slot].synchronize()
        for i, (vals, dst) in enumerate(zip(rows, dsts)):
            if cuda:
                stage = self._hyper_ring[self._hyper_slot, i]
                stage[: len(vals)] = torch.tensor(vals, dtype=torch.float32)
                dst[: len(vals)].copy_(stage[: len(vals)], non_blocking=True)
            else:
                dst[: len(vals)].copy_(torch.tensor(vals, dtype=torch.float32))
        if cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_done[self._hyper_slot] = ev
            self._hyper_slot = (self._hyper_slot + 1) % self._hyper_ring.shape[0]

    # -------------------------------------------------------------------------------- the step body
    # The step is cut into three phases so that each CAN be its own hipGraph (TextStack(concurrent=True)): ROCm's graph
    # executor only overlaps forked branches (the two text encoders) when the fork sits near the root of a small graph -
    # inside the whole-step graph the branches were replayed strictly one after the other (kernel trace,
    # tools/graph_branch_probe2.py).  By default the phases are captured back to back into ONE graph.
    def _phase_text_fwd(self):
        self._pooled_live = self.pooled
        if self.text is not None:                      # a4: text conditioning with the trainable token rows (main.py:306-308)
            p = self.text.forward(self.ctx)
            self._pooled_live = p if p is not None else self.pooled

    def _phase_unet(self):
        rt, u = self.rt, self.unet
        ops = rt.ops
        ops.add_noise_nhwc(self.latent, self.noise, self.timesteps, self.acp, self.x64, self.noisy)
        pred = u.forward(self.x64, self.timesteps_f, self.ctx, self._pooled_live, self.time_ids, B=self.B, H=self.h, W=self.w)
        scale = 1.0 / self.grad_accum
        ops.masked_mse_fwd_bwd(pred, self.noise, self.noisy, self.mask, self.timesteps, self.acp, self.sums, self.loss,
                               self.dpred64, snr_gamma=self.snr_gamma, v_prediction=self.v_pred, loss_scale=scale)
        rt.daam_grads = None
        if self.text is not None and self.ta_w > 0.0:  # a10/a11: token-attention loss on the hooked score maps (main.py:342-345)
            self.ta.forward_backward(self.mask, self.w / self.h, self.ta_w * scale)
        rt.daam_applied = False
        u.daam_backward()
        self.dctx.zero_()
        u.backward(self.dpred64, self.dctx)
        self._pred = pred

    def _phase_text_bwd(self):
        if self.text is not None:
            P = self.pooled.shape[1] if self.pooled is not None else 0
            d_pooled = self.unet.dadd_in[:, :P] if self.rt.want_dpooled else None
            if self.cond_reg_w > 0.0:
                # cond_reg_w * (mean_{t >= 2} mean_b |prompt_embeds[b, t]| - target)^2  (loss.py:201-205, 235-239): its gradient
                # joins the UNet's gradient w.r.t. the conditioning before the text-encoder backward
                pe = self.ctx.view(self.B, CTX_PAD, -1)[:, 2:T_TOKENS].float()
                nrm = pe.norm(dim=-1, keepdim=True)
                val = nrm.mean()
                self.cond_norm.copy_(val.reshape(1))
                self.cond_reg_loss.copy_((self.cond_reg_w * (val - self.cond_target_norm) ** 2).reshape(1))
                coef = self.cond_reg_w / self.grad_accum * 2.0 * (val - self.cond_target_norm) / (self.B * (T_TOKENS - 2))
                dv = self.dctx.view(self.B, CTX_PAD, -1)
                dv[:, 2:T_TOKENS] += (coef * pe / nrm).to(dv.dtype)
            self.text.backward(self.dctx, d_pooled, self.ti.grad_rows)
            if self.tok_cond_reg_w > 0.0:
                self._tok_cond_reg()
            if self.tok_cov_reg_w > 0.0:
                self.ti.add_covariance(self.tok_cov_reg_w / self.grad_accum)
            # a14 (only the std term is live by default, config.py:75-77); part of the loss, hence / k under accumulation
            self.ti.add_regulariser(std_loss_w=self.ti.std_loss_w / self.grad_accum)

    def _tok_cond_reg(self):
        """loss += tok_cond_reg_w * (mean_{t >= 2} mean_c |E(caption_c)[t]| - target)^2 over the regularisation captions
        (ConditioningRegularizer._compute_tok_regularization_loss, loss.py:241-251); back through the encoders into the rows."""
        nreg, off = self.reg_rt.B, 0
        for e, ids, w in zip(self.reg_encoders, self.reg_ids, self.text.widths):
            e.forward(ids, nreg, hidden_out=self.reg_ctx[:, off:off + w], hidden_only=True)
            off += w
        pe = self.reg_ctx.view(nreg, CTX_PAD, -1)[:, 2:T_TOKENS].float()
        nrm = pe.norm(dim=-1, keepdim=True)
        val = nrm.mean()
        self.tok_reg_norm.copy_(val.reshape(1))
        self.tok_reg_loss.copy_((self.tok_cond_reg_w * (val - self.cond_target_norm) ** 2).reshape(1))
        coef = self.tok_cond_reg_w / self.grad_accum * 2.0 * (val - self.cond_target_norm) / (nreg * (T_TOKENS - 2))
        self.reg_dctx.zero_()
        self.reg_dctx.view(nreg, CTX_PAD, -1)[:, 2:T_TOKENS] = (coef * pe / nrm).to(self.reg_dctx.dtype)
        off = 0
        for e, w, g in zip(self.reg_encoders, self.text.widths, self.ti.grad_rows):
            e.backward(self.reg_dctx[:, off:off + w], None, g, accumulate=True)
            off += w
        if self.reg_arena is not None:
            if self._reg_grad_plan is None:
                self._reg_grad_plan = self.rt.ops.LoraGradPlan(self.reg_arena.problems, self.reg_arena.Rp, self.rt.device)
                self._reg_grad_plan.set_accumulate(True)
            self._reg_grad_plan.run()
            if self.text.arena.dora:
                self.text.arena.dora_mag_grad()          # first pass: its magnitude gradients, and the column factor onto the summed dB rows
                if self._reg_dora_plan is None:
                    grads = [dict(g, accumulate=True) for g in self.reg_arena.dora_grads]
                    self._reg_dora_plan = self.rt.ops.DoraPlan([], [], grads, self.text.arena.rank, self.text.arena.Rp, self.rt.device)
                self._reg_dora_plan.mag_grad()

    def forward_backward(self):
        self._phase_text_fwd()
        self._phase_unet()
        self._phase_text_bwd()
        return self._pred

    def _unet_optimizer(self):
        a = self.group
        l1 = None if self.full_ft else self.l1_sum
        if self.full_ft and self.prodigy is None:
            a.adamw_step(self.hyper)           # optimizer step and operand refresh in one tiled pass over the matrices
            return
        if self.prodigy is not None:
            # Prodigy under data parallelism: its step-size estimate d is built from sums of g . (p0 - p) and |s| and is not invariant to the
            # gradient's scale, so the SUMMED gradients are turned into the mean (one in-place multiply) before its two passes - the state every
            # rank then holds is the state of one process on the whole batch; AdamW takes the mean through its hyper row instead (no extra pass).
            # (Text-encoder LoRA: the adapter gradients are exchanged like the token rows - the ranks see different captions.)
            if self.world > 1:
                a.grads.mul_(1.0 / self.world)
            self.prodigy.step(a.grads, a.m, a.v, self.hyper, l1)
        else:
            self.rt.ops.adamw_fused(a.params, a.grads, a.m, a.v, self.hyper, l1)
        a.refresh_shadows()

    def sync_gradients(self):
        """Data-parallel full fine-tune, unbucketed form (gradient accumulation): ONE all-reduce (sum) of the flat fp32 gradient
        arena per optimiser step.  RCCL over xGMI on the GPU (backend "nccl"), gloo in the CPU tests."""
        if self.ddp and not getattr(self, "bucketed", False):
            import torch.distributed as dist
            self._log("all_reduce", self.group.grads, self.group.grads, self.group.grads)
            dist.all_reduce(self.group.grads, group=self.pg)
            if self.ti is not None:
                dist.all_reduce(self.ti.grads, group=self.pg)
            if self.te_arena is not None:
                dist.all_reduce(self.te_arena.grads, group=self.pg)

    def _log(self, op, base, inp, out):
        """Record a collective for the call-sequence checks (tests/test_parallel_cpu.py, bench.py --dry-collectives): element offsets of the input / output views inside
        `base` (the arena they are slices of), their sizes and dtype."""
        if self.coll_log is not None:
            es = base.element_size()
            self.coll_log.append((op, (inp.data_ptr() - base.data_ptr()) // es, inp.numel(), (out.data_ptr() - base.data_ptr()) // es, out.numel(), str(inp.dtype).replace("torch.", ""),
                                  inp.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0))

    def flush_and_reduce(self, flush_fns=None):
        """Data-parallel full fine-tune, the exchange step of the path: the deferred weight-gradient plan runs bucket
        by bucket (contiguous >= 256 MB ranges of the fp32 gradient arena, fullft.WeightTrainer.buckets); after bucket i's GEMMs are
        queued its all-reduce (sum, in place) is launched asynchronously - torch.distributed makes the collective's stream wait for
        the work queued so far and runs it beside bucket i+1's GEMMs - then the small vector region (biases, norm affine), then all
        are waited for before the optimizer.  flush_fns: per-bucket callables (the captured hipGraphs' replays); default eager."""
        import torch.distributed as dist
        tr = self.group
        works = []
        if self.ti is not None:      # token-row gradients (a few KB): complete after the backward graph, exchanged beside the first bucket
            self._log("all_reduce:ti", self.ti.grads, self.ti.grads, self.ti.grads)
            works.append((dist.all_reduce(self.ti.grads, group=self.pg, async_op=True), None, None))
        if self.te_arena is not None:    # text-encoder adapter gradients (a few MB), complete after the backward graph as well
            self._log("all_reduce:te", self.te_arena.grads, self.te_arena.grads, self.te_arena.grads)
            works.append((dist.all_reduce(self.te_arena.grads, group=self.pg, async_op=True), None, None))
        wire = getattr(self, "wire", None)
        zero1 = getattr(self, "zero1", False)
        for b, (o0, o1) in enumerate(tr.buckets):
            (flush_fns[b] if flush_fns is not None else (lambda b=b: tr.flush(bucket=b)))()
            if zero1:
                # reduce-scatter IN PLACE (output = this rank's slice of the input: the form RCCL runs without a staging copy): after it
                # grads[s0:s1] holds the sum over the ranks, the rest of the bucket is scratch until the next backward overwrites it
                s0, s1 = tr.shard_range(b)
                if wire is not None:
                    wire[o0:o1].copy_(tr.grads[o0:o1])       # pack: fp32 -> bf16; only the owned slice is unpacked afterwards
                    self._log("reduce_scatter", wire, wire[o0:o1], wire[s0:s1])
                    works.append((dist.reduce_scatter_tensor(wire[s0:s1], wire[o0:o1], group=self.pg, async_op=True), s0, s1))
                else:
                    self._log("reduce_scatter", tr.grads, tr.grads[o0:o1], tr.grads[s0:s1])
                    works.append((dist.reduce_scatter_tensor(tr.grads[s0:s1], tr.grads[o0:o1], group=self.pg, async_op=True), None, None))
            elif wire is not None:
                o1w = min(o1, tr.n_mat)
                wire[o0:o1w].copy_(tr.grads[o0:o1w])           # pack: fp32 -> bf16, queued behind the bucket's GEMMs
                self._log("all_reduce", wire, wire[o0:o1w], wire[o0:o1w])
                works.append((dist.all_reduce(wire[o0:o1w], group=self.pg, async_op=True), o0, o1w))
            else:
                self._log("all_reduce", tr.grads, tr.grads[o0:o1], tr.grads[o0:o1])
                works.append((dist.all_reduce(tr.grads[o0:o1], group=self.pg, async_op=True), None, None))
        if tr.n > tr.n_mat:
            self._log("all_reduce:vec", tr.grads, tr.grads[tr.n_mat:], tr.grads[tr.n_mat:])
            works.append((dist.all_reduce(tr.grads[tr.n_mat:], group=self.pg, async_op=True), None, None))
        for w, o0, o1w in works:
            w.wait()
            if o0 is not None:
                tr.grads[o0:o1w].copy_(wire[o0:o1w])            # unpack: the optimizer reads fp32 sums

    def gather_params(self):
        """ZeRO-1, after the sharded AdamW: every bucket's updated master slices -> all ranks (all-gather in place: the input is this rank's slice
        of the output), outside the hipGraphs like every collective of the path."""
        import torch.distributed as dist
        tr = self.group
        works = []
        for b, (o0, o1) in enumerate(tr.buckets):
            s0, s1 = tr.shard_range(b)
            self._log("all_gather", tr.params, tr.params[s0:s1], tr.params[o0:o1])
            works.append(dist.all_gather_into_tensor(tr.params[o0:o1], tr.params[s0:s1], group=self.pg, async_op=True))
        for w in works:
            w.wait()

    def _opt_shard_phase(self):
        """ZeRO-1 optimizer, first half (one hipGraph): AdamW over the owned slices + every replicated optimizer of the step."""
        self.group.adamw_shard_step(self.hyper)
        self._other_optimizers()

    def _opt_post_phase(self):
        """ZeRO-1 optimizer, second half (one hipGraph, after gather_params): fp32 masters -> the bf16 W / W^T operands of every matrix."""
        self.group.refresh()

    def optimizer_step(self):
        if getattr(self, "zero1", False):
            self._opt_shard_phase()
            self.gather_params()
            self._opt_post_phase()
            return
        self._unet_optimizer()
        self._other_optimizers()

    def _other_optimizers(self):
        if self.ti is not None:                        # a17: the trainable token rows only
            t = self.ti
            if self.prodigy_ti is not None:
                if self.world > 1:
                    t.grads.mul_(1.0 / self.world)
                self.prodigy_ti.step(t.grads, t.m, t.v, t.hyper, None)
            else:
                self.rt.ops.adamw_fused(t.params, t.grads, t.m, t.v, t.hyper, None)
            t.refresh_tables()
        if self.te_arena is not None:                  # a21, between TI and UNet in the reference's order (optimizer.py:265-275)
            e = self.te_arena
            self.rt.ops.adamw_fused(e.params, e.grads, e.m, e.v, self.te_hyper, None)
            e.refresh_shadows()

    def _accumulate(self, release):
        if self._acc is not None:
            for g_, a in self._acc:
                a.add_(g_)
                if release:
                    g_.copy_(a)
                    a.zero_()

    def body_micro(self):
        """A micro-step that is not an accumulation boundary: gradients only."""
        self.forward_backward()
        self._accumulate(False)

    def body(self):
        self.forward_backward()
        if getattr(self, "bucketed", False):
            self.flush_and_reduce()
        else:
            self._accumulate(True)
            self.sync_gradients()
        self.optimizer_step()

    def _phases(self):
        if self.ddp and self.bucketed:   # forward+backward | per bucket: weight gradients, exchange (outside the graphs) | optimizer
            tr = self.group
            opt = [self._opt_shard_phase, self._opt_post_phase] if getattr(self, "zero1", False) else [self.optimizer_step]     # (the all-gather sits between the two)
            return [self.forward_backward] + [(lambda b=b: tr.flush(bucket=b)) for b in range(len(tr.buckets))] + opt
        if self.ddp:         # the collective stays outside the graphs: forward+backward | all-reduce | optimizer
            return [lambda: (self.forward_backward(), self._accumulate(True)), self.optimizer_step]
        if self._acc is not None:
            return [self.body]
        if self.text is None:
            return [self.body]
        return [self._phase_text_fwd, self._phase_unet, lambda: (self._phase_text_bwd(), self.optimizer_step())]

    def _phase_unet_cached(self):
        """UNet phase on a conditioning the caller placed in self.ctx / self.pooled (frozen token rows: no text-encoder forward)."""
        self._pooled_live = self.pooled
        self._phase_unet()

    def _phase_opt_frozen_ti(self):
        """Last phase once the token embeddings are frozen (ti lr == 0, main.py:273-274): the reference still back-propagates
        through both text encoders and runs AdamW with lr 0 on the tables; with lr == 0 that changes no parameter (decoupled
        decay is lr * wd) and the regulariser is skipped (main.py:358), so only the LoRA optimiser remains."""
        self._unet_optimizer()

    # -------------------------------------------------------------------------------- a20: token warm-up
    def token_warmup(self, prompt_ids, target_ids, steps, lr, weight_decay=None):
        """`pre_optimize_token_embeddings` (trainer/embedding_handler.py:321-399; off unless token_warmup_steps > 0 and a
        gpt_description exists): optimise the token rows WITHOUT the denoiser so that the encoding of the bare trigger
        prompt moves towards the encoding of a description.  Per step: encode the trigger prompt, loss = 0.2 * (MSE +
        1 - cos [+ 0.25 * the same on the pooled embedding]) against the detached target (:288-318) + 0.5 * the token-std
        regulariser (:384), back through the text encoders to the token rows, AdamW (a FRESH optimizer: its moments are
        discarded afterwards, :345-354).  prompt_ids / target_ids: one int64 [77] row per tokenizer.  Runs eagerly (a
        one-off loop before training); the objective's few hundred KB of arithmetic are torch ops, everything else is
        the text-encoder plan of the step.  Returns the per-step loss values."""
        text, ti, rt, B = self.text, self.ti, self.rt, self.B
        dev = rt.device
        ctxv = self.ctx.view(B, CTX_PAD, -1)

        def encode(ids_per_tok):
            text.set_ids([i.to(dev).view(1, T_TOKENS).expand(B, T_TOKENS) for i in ids_per_tok])
            return text.forward(self.ctx)

        pooled = encode(target_ids)                     # the target is encoded with the CURRENT rows and then detached
        tgt = ctxv[:, :T_TOKENS].float().clone()
        tgt_pooled = pooled.float().clone() if pooled is not None else None
        tgt_n = tgt.norm(dim=-1, keepdim=True)
        m, v = torch.zeros_like(ti.params), torch.zeros_like(ti.params)
        wd = self.ti_wd if weight_decay is None else weight_decay
        b1, b2 = self.betas
        n_rows, D = B * T_TOKENS, tgt.shape[-1]
        d_pooled = rt.zeros(*pooled.shape) if pooled is not None else None
        losses = []
        for k in range(1, steps + 1):
            pooled = encode(prompt_ids)
            p = ctxv[:, :T_TOKENS].float()
            pn = p.norm(dim=-1, keepdim=True)
            cos = (p * tgt).sum(-1, keepdim=True) / (pn * tgt_n)
            loss = ((p - tgt) ** 2).mean() + 1.0 - cos.mean()
            dp = 2.0 * (p - tgt) / (n_rows * D) - (tgt / (pn * tgt_n) - cos * p / (pn * pn)) / n_rows
            if pooled is not None:
                q = pooled.float()
                qn, tn = q.norm(dim=-1, keepdim=True), tgt_pooled.norm(dim=-1, keepdim=True)
                cq = (q * tgt_pooled).sum(-1, keepdim=True) / (qn * tn)
                loss = loss + 0.25 * (((q - tgt_pooled) ** 2).mean() + 1.0 - cq.mean())
                dq = 2.0 * (q - tgt_pooled) / q.numel() - (tgt_pooled / (qn * tn) - cq * q / (qn * qn)) / B
                d_pooled.copy_(0.2 * 0.25 * dq)
            self.dctx.zero_()
            self.dctx.view(B, CTX_PAD, -1)[:, :T_TOKENS] = (0.2 * dp).to(self.dctx.dtype)
            text.backward(self.dctx, d_pooled, ti.grad_rows)
            ti.add_regulariser(std_loss_w=0.5)
            ti.hyper[:9] = torch.tensor([lr, b1, b2, self.eps, wd, 1.0 - b1 ** k, 1.0 - b2 ** k, 0.0, 1.0], dtype=F32)
            rt.ops.adamw_fused(ti.params, ti.grads, m, v, ti.hyper, None)
            ti.refresh_tables()
            losses.append(0.2 * float(loss) + float(ti.reg_loss))
        return losses

    def grad_norm(self):
        """Global L2 norm of the LoRA gradients, the reference's debug read-out (loss.py:108-125, main.py:373-379).
        COLLECTIVE under the sharded data-parallel optimizer (ZeRO-1): every rank must call it (one scalar all-reduce) - a rank-0-only call, the way
        the reference guards its logging, would hang the job; with the all-reduce exchange or on one GPU it is a local read."""
        return self._unet_grad_norm()

    def _unet_grad_norm(self):
        """L2 norm of the UNet gradient the optimizer consumes.  Data parallel: the gradient of the global batch is the rank sum / world.  After
        the all-reduce every rank holds the sum; after ZeRO-1's in-place reduce-scatter only this rank's slice of each bucket does (the rest of
        the bucket is scratch), so the squared norm is taken over the owned slices + the replicated vector region once (rank 0) and the
        scalar is summed over the ranks."""
        a = self.group
        if not self.ddp:
            return float(a.grads.norm())
        if getattr(self, "zero1", False):
            import torch.distributed as dist
            sq = torch.zeros(1, dtype=torch.float64, device=a.grads.device)
            for b in range(len(a.buckets)):
                s0, s1 = a.shard_range(b)
                sq += a.grads[s0:s1].double().pow(2).sum()
            if a.n > a.n_mat and a.z_rank == 0:
                sq += a.grads[a.n_mat:].double().pow(2).sum()
            dist.all_reduce(sq, group=self.pg)
            return math.sqrt(float(sq)) / self.world
        return float(a.grads.norm()) / self.world

    def grad_norms(self):
        """The whole debug read-out of main.py:373-379: {'unet': ..., 'text_encoder_0': ..., 'text_encoder_1': ...} - `compute_grad_norm` over
        every parameter that has a gradient.  For a text encoder that is its token table AFTER the rows of the frozen vocabulary were zeroed
        (main.py:368-371: only the trained rows count - this engine never forms the other rows' gradients) plus, with
        text_encoder_lora_optimizer, that encoder's adapters.  One device sync per entry; not part of the step's graph.  Like grad_norm(): a
        collective under ZeRO-1 - call it on every rank or on none."""
        out = {"unet": self._unet_grad_norm()}
        if self.ti is not None:
            sq = [float(r.float().pow(2).sum()) for r in self.ti.grad_rows]
            if self.te_arena is not None:
                for e in self.te_arena.entries:
                    i = 1 if e["name"].startswith("text_encoder_2.") else 0
                    sq[i] += float(e["gA"].float().pow(2).sum()) + float(e["gB"].float().pow(2).sum())
            for i, v in enumerate(sq):
                out[f"text_encoder_{i}"] = math.sqrt(v)
        return out

    # -------------------------------------------------------------------------------- graph capture / replay
    def _ws(self):
        """Scope in which this job's GEMMs use their own split-K workspace (ops.workspace_owner); a no-op for the CPU emulation."""
        own = getattr(self.rt.ops, "workspace_owner", None)
        if own is None:
            import contextlib
            return contextlib.nullcontext()
        if self.text is not None and self.text.concurrent:      # forked encoders keep one split-K workspace per stream; the norm scratch is per job
            return self.rt.ops.norm_workspace_owner(id(self))
        return own(id(self))

    def capture(self, warmup=2):
        with self._ws():
            return self._capture(warmup)

    def _capture(self, warmup=2):
        """Runs the body eagerly `warmup` times (allocates every persistent buffer, builds the grouped-gradient
        plan), then captures it: one hipGraph for the whole step (one per phase when the text encoders run on forked
        streams), plus the frozen-TI variant.  AdamW state is restored afterwards so capture does not count as training."""
        a = self.group
        state = [a.params] + (a.opt_state() if hasattr(a, "opt_state") else [a.m, a.v]) + ([self.ti.params, self.ti.m, self.ti.v] if self.ti is not None else [])
        if self.te_arena is not None:
            state += [self.te_arena.params, self.te_arena.m, self.te_arena.v]
        snap = [t.clone() for t in state]
        step0 = self.opt_step
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.body()
        torch.cuda.current_stream().wait_stream(s)
        ops = self.rt.ops
        prefetch = hasattr(ops, "pf_record_begin") and getattr(ops, "WSK_PREFETCH", False) and not self.ddp and not self.full_ft      # (frozen, packed weights only: LoRA / TI jobs)

        def cap(fns, pool):
            # next-weight prefetch (ops.pf_*): one more eager pass of exactly these phases records the sequence of wave-split-K products, the capture replays it so that every
            # such launch knows the packed weight of the one behind it (the optimizer state this pass moves is restored below, like the warm-up's)
            seq = None
            if prefetch:
                with torch.cuda.stream(s):
                    ops.pf_record_begin()
                    try:
                        for fn in fns:
                            fn()
                    finally:
                        seq = ops.pf_record_end()
                torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                if seq:
                    ops.pf_replay_begin(seq)
                try:
                    for fn in fns:
                        fn()
                finally:
                    if seq:
                        ops.pf_replay_end()
            return g
        phases = self._phases()
        split = (self.text is not None and self.text.concurrent) or self.ddp      # one graph per phase when the encoders fork / DDP

        def cap_set(pool):
            graphs = []
            for fns in ([[ph] for ph in phases] if split else [phases]):
                graphs.append(cap(fns, pool))
                pool = graphs[-1].pool()
            frozen = None
            # variant for ti lr == 0: same first phases, LoRA-only last phase.  Not under data parallelism: _run never takes the frozen branch there
            # (the exchange step lives in the per-bucket graphs), and with ZeRO-1 the full-arena AdamW those graphs would record has no moments
            if self.text is not None and self.te_arena is None and not self.ddp:
                if split:
                    frozen = graphs[:2] + [cap([self._phase_opt_frozen_ti], pool)]
                else:
                    frozen = [cap([self._phase_text_fwd, self._phase_unet, self._phase_opt_frozen_ti], pool)]
                pool = frozen[-1].pool()
                self.graphs_frozen_cached = [cap([self._phase_unet_cached, self._phase_opt_frozen_ti], pool)]   # + cached conditioning
            return graphs, frozen, pool

        self.graphs, self.graphs_frozen, pool = cap_set(None)
        self.graph = self.graphs[0]
        if self._acc is not None:
            self.graphs_frozen = None
            self.graphs_micro = [cap([self.body_micro], pool)]
            for _, acc_buf in self._acc:
                acc_buf.zero_()
        for t, c in zip(state, snap):
            t.copy_(c)
        if not self.full_ft:
            # the logged sum of |p| (a read-out the optimizer kernel leaves; the capture passes ran it on parameters that are restored above - with more than one eager pass on a
            # zero hyper row they are not even finite): what a gradient-accumulation micro-step logs before the first optimizer step is the restored parameters' own sum
            self.l1_sum.copy_(a.params.abs().sum().reshape(1))
        a.refresh_shadows()
        if self.ti is not None:
            self.ti.refresh_tables()
        if self.te_arena is not None:
            self.te_arena.refresh_shadows()
        for pr in (self.prodigy, self.prodigy_ti):
            if pr is not None:
                pr.reset()
        self.opt_step = step0

    def run(self, lr, lr_ti=0.0, lr_te=0.0, last_batch=False):
        with self._ws():
            return self._run(lr, lr_ti, lr_te, last_batch)

    def _run(self, lr, lr_ti=0.0, lr_te=0.0, last_batch=False):
        if self._acc is not None:
            self._micro += 1
            if self._micro % self.grad_accum != 0 and not last_batch:        # main.py:366
                if self.graphs_micro is not None:
                    self.graphs_micro[0].replay()
                else:
                    self.body_micro()
                return
            self._micro = 0
        if not self.ti_trainable:
            lr_ti = 0.0
        self.set_hyper(lr, lr_ti, lr_te)
        # frozen-TI fast path (Prodigy with lr 0 is a no-op as well: sdlt_prodigy_step); never with text-encoder LoRA, whose
        # gradients need the text backward for the whole run
        # (not under data parallelism: the exchange step lives in body() / the per-bucket graphs, and the eager frozen branch below would skip it)
        frozen = self.text is not None and lr_ti == 0.0 and self.te_arena is None and self._acc is None and not self.ddp
        self._frozen_last = frozen
        if self.graph is not None and self.ddp and self.bucketed and getattr(self, "zero1", False):
            self.graphs[0].replay()
            self.flush_and_reduce([g.replay for g in self.graphs[1:-2]])
            self.graphs[-2].replay()
            self.gather_params()
            self.graphs[-1].replay()
        elif self.graph is not None and self.ddp and self.bucketed:
            self.graphs[0].replay()
            self.flush_and_reduce([g.replay for g in self.graphs[1:-1]])
            self.graphs[-1].replay()
        elif self.graph is not None and self.ddp:
            self.graphs[0].replay()
            self.sync_gradients()
            self.graphs[1].replay()
        elif self.graph is not None:
            cached = frozen and self._cond_cached and self.graphs_frozen_cached is not None
            assert not self._cond_cached or frozen, "a cached conditioning is only valid while the token rows are frozen (ti lr == 0, no text-encoder LoRA)"
            for g in (self.graphs_frozen_cached if cached else (self.graphs_frozen if frozen else self.graphs)):
                g.replay()
        elif frozen:
            if self._cond_cached:
                self._phase_unet_cached()
            else:
                self._phase_text_fwd()
                self._phase_unet()
            self._phase_opt_frozen_ti()
        else:
            self.body()

    def total_loss(self):
        """img loss + token-attention loss + L1 penalty + the token regularisers as the reference logs it (main.py:339-361);
        forces a device sync."""
        tot = float(self.loss) + self.l1_penalty * float(self.l1_sum) / self.group.n
        if self.text is not None:
            tot += self.ta_w * float(self.ta.loss)
            if not getattr(self, "_frozen_last", False) and self.ti_trainable:       # main.py:358: the regularisers only while the TI learning rate is > 0
                tot += float(self.ti.reg_loss)
                if self.cond_reg_w > 0.0:
                    tot += float(self.cond_reg_loss)
                if self.tok_cond_reg_w > 0.0:
                    tot += float(self.tok_reg_loss)
                if self.tok_cov_reg_w > 0.0:
                    tot += float(self.ti.cov_loss)
        return tot
