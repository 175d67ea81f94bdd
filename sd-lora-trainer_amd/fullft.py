"""Full-UNet fine-tune: `is_lora = False` in the reference means `unet.requires_grad_(True)` and an
optimizer over `unet.parameters()` (/root/reference main.py:144-149, trainer/optimizer.py:6-39); the example config is
train_configs/full_finetuning_example.json (SDXL, 512 px, batch 4, no textual inversion).

`WeightTrainer` is the flat fp32 master copy of EVERY UNet parameter (weights, biases, norm affine) with its gradient and
AdamW moments, the weight-gradient plan the leaf layers call from their backward, and the refresh of the bf16 compute
copies (both orientations) after each optimizer step.

  * dW[N,K] = dY^T X contracts over tokens: both operands are transposed once into token-contiguous panels
    (sdlt_wgrad_transpose / sdlt_wgrad_im2col_t) and multiplied by the MFMA GEMM with an fp32 output that IS the
    parameter's slice of the gradient arena (no per-parameter gradient tensors, one all-reduce buffer for data parallel).
  * biases: column sums of dY, a by-product of the pass that transposes dY; norm affine: sdlt_*_affine_grad.  All vector
    gradients live in one tail region of the arena, cleared by one launch per step.  Biases and gamma/beta are used by the kernels in fp32,
    so the layers read them straight from the master arena - only matrix weights have bf16 copies to refresh.
  * 3x3 conv weights are stored tap-major [Cout, (ky, kx, ci)] like the forward GEMM operand; `export()` / `load()`
    convert to / from the PyTorch [Cout, Cin, 3, 3] layout of the checkpoint.
"""
import torch

F32, BF16 = torch.float32, torch.bfloat16


def _pad64(n):
    return (n + 63) // 64 * 64


class WeightTrainer:
    def __init__(self, rt):
        self.rt = rt
        self.entries = []          # dict(name, off, shape (arena layout), kind)
        self.by_name = {}
        self.n = 0                 # matrix region [0, n_mat), then the vector region (biases, norm affine) [n_mat, n)
        self.nv = 0
        self._binds = []           # callables run at finalize (point layer attributes at arena views)
        self._shadow = []          # ShadowPlan descriptors (offset, rows, cols, src_ld, dst, dstT)
        self.params = self.grads = self.m = self.v = None
        self.q8 = None             # AdamW8bit: (m8, v8, absmax, tables), see enable_8bit
        self.q8_sh = None          # AdamW8bit under ZeRO-1: ([absmax of this rank's slice of bucket b], tables), see enable_zero1
        self.registering = False   # True only while the UNet builds its layers (the text encoders share the leaf classes)
        # Deferred, batched weight gradients: the leaf layers only RECORD (weight, inputs, dY) during the first backward; flush()
        # at the end of every backward issues all layers of one shape together - one panel launch per operand and one batched
        # GEMM for e.g. the 60 FF projections of the 1280-wide transformer blocks.  Legal because every activation and every
        # layer gradient lives in a persistent, layer-owned buffer until the next forward.
        self.defer = True
        self._jobs, self._wplan = [], None
        self._affine_jobs, self._aplan = [], None
        # Data parallel: the matrix region of the arena is laid out CLASS BY CLASS (kind, shape) - the unit the deferred
        # weight-gradient plan works in - and cut into contiguous buckets of >= bucket_floats; flush(bucket=i) produces exactly
        # the gradients of bucket i, so its all-reduce can be launched (in place, on the collective's stream) while the next
        # bucket's weight-gradient GEMMs run (step.TrainStep with a process group).
        self.bucket_floats = 64 << 20          # 256 MB of fp32 gradients per bucket
        self.buckets = []                      # [(off0, off1)] over the matrix region, in flush order
        self.defer_flush = False               # True: UNet.backward leaves the flush to the caller (bucket by bucket)

    # ------------------------------------------------------------------ registration (layer constructors)
    def add(self, name, init, kind="matrix"):
        """init: fp32 tensor in ARENA layout.  Offsets are kept 16-byte aligned so every slice can be a GEMM output."""
        assert self.params is None, "register before finalize()"
        if kind == "vector":
            # vectors live behind all matrices: their gradients are accumulated with atomics (column sums, norm affine), so the
            # whole region is cleared with ONE launch per step instead of one per tensor
            self.nv = (self.nv + 3) // 4 * 4
            e = dict(name=name, voff=self.nv, off=None, shape=tuple(init.shape), kind=kind, init=init.detach().to(F32))
            self.nv += init.numel()
        else:               # the offset is assigned at finalize(): matrices are laid out class by class (see buckets)
            e = dict(name=name, off=None, seq=len(self.entries), shape=tuple(init.shape), kind=kind, init=init.detach().to(F32))
        self.entries.append(e)
        self.by_name[name] = e
        return e

    def on_finalize(self, fn):
        self._binds.append(fn)

    def shadow(self, entry, rows, cols, src_ld, dst, dstT, offset=0):
        self._shadow.append((entry, offset, rows, cols, src_ld, dst, dstT))

    def finalize(self):
        rt = self.rt
        # matrix region: classes (kind, shape) in order of first registration, members in registration order (stacked q|k|v
        # projections are registered back to back with one shape, so they stay adjacent); buckets = runs of whole classes
        classes = {}
        for e in self.entries:
            if e["kind"] != "vector":
                # (a Linear and a 1x1 conv of one shape are the same kind of weight-gradient job and may share a flush group)
                classes.setdefault(("conv3x3" if e["kind"] == "conv3x3" else "mat", e["shape"]), []).append(e)
        self.n, start = 0, 0
        for members in classes.values():
            for e in members:
                self.n = (self.n + 3) // 4 * 4
                e["off"] = self.n
                self.n += int(torch.tensor(e["shape"]).prod())
            if self.n - start >= self.bucket_floats:
                # bucket ends sit on multiples of 64 floats: every bucket splits into equal, 16-byte aligned shards for 2 / 4 / 8 / 16 ranks
                # (reduce-scatter / all-gather of the sharded optimizer, enable_zero1); the padding belongs to no tensor and stays zero
                self.n = (self.n + 63) // 64 * 64
                self.buckets.append((start, self.n))
                start = self.n
        self.n_mat = (self.n + 63) // 64 * 64
        if self.n_mat > start:
            self.buckets.append((start, self.n_mat))
        for e in self.entries:
            if e["kind"] == "vector":
                e["off"] = self.n_mat + e.pop("voff")
        self.n = self.n_mat + self.nv
        z = lambda: torch.zeros(self.n, dtype=F32, device=rt.device)  # noqa: E731
        self.params, self.grads, self.m, self.v = z(), z(), z(), z()
        for e in self.entries:
            self.view(e).copy_(e.pop("init").to(rt.device))
        for fn in self._binds:
            fn()
        sh = [(e["off"] + o, r, c, ld, d, dt) for (e, o, r, c, ld, d, dt) in self._shadow]
        self._plan = rt.ops.ShadowPlan(sh, rt.device) if sh else None

    def view(self, e, which="params"):
        t = getattr(self, which)
        n = 1
        for s in e["shape"]:
            n *= s
        return t[e["off"]: e["off"] + n].view(e["shape"])

    def refresh(self):
        """fp32 master -> bf16 compute copies (W and W^T of every matrix / conv weight)."""
        if self._plan is not None:
            self._plan.run(self.params)

    def adamw_step(self, hyper):
        """torch.optim.AdamW over the whole arena and the refresh of the bf16 operands: the matrix region in ONE tiled pass
        (sdlt_adamw_shadow_refresh: p, g, m, v -> p, m, v, W, W^T), the vector region (biases, norm affine: used in fp32) with
        sdlt_adamw_fused."""
        ops, nm = self.rt.ops, self.n_mat
        if self.q8 is not None:           # AdamW8bit: byte moments for the matrices, fp32 moments for the vector region
            m8, v8, absmax, tables = self.q8
            self._plan.adamw8(self.params[:nm], self.grads[:nm], m8, v8, absmax, tables, hyper)
            if self.nv:
                ops.adamw_fused(self.params[nm:], self.grads[nm:], self.m_vec, self.v_vec, hyper, None)
            return
        self._plan.adamw(self.params, self.grads, self.m, self.v, hyper)
        if self.nv:
            ops.adamw_fused(self.params[nm:], self.grads[nm:], self.m[nm:], self.v[nm:], hyper, None)

    def enable_8bit(self):
        """`unet_optimizer_type: AdamW8bit` (trainer/optimizer.py:19-21, full_finetuning_example.json): the moments of every matrix / conv weight as one byte per
        element + one fp32 absmax per 2048-element block (include/sdlt_kernels.h: sdlt_adamw8_shadow_refresh) - 2 x 10.3 GB of fp32 moments become 2 x 2.6 GB on
        SDXL and the optimizer pass moves 20 instead of 32 bytes per parameter.  The vector region (biases, norm affine parameters: 0.1 % of the arena) keeps fp32
        moments: bitsandbytes does the same for every tensor under 4096 elements, and quantises the 140 GEGLU / conv biases above that size, which stay fp32 here."""
        if getattr(self, "q8", None) is not None:          # (idempotent: a second TrainStep on the same trainer finds the 8-bit state in place)
            return
        assert self.params is not None and self.m is not None and self._plan is not None
        nm, dev = self.n_mat, self.rt.device
        self.m_vec, self.v_vec = self.m[nm:].clone(), self.v[nm:].clone()
        self.m = self.v = None
        nb = self._plan.n_blocks          # codes are tile-major: 4096 per 64 x 64 tile of the refresh plan (ragged tiles leave positions unused)
        self.q8 = (torch.zeros(4096 * nb, dtype=torch.uint8, device=dev), torch.zeros(4096 * nb, dtype=torch.uint8, device=dev),
                   torch.zeros(4 * nb, dtype=F32, device=dev), self.rt.ops.q8_tables(dev))

    # ------------------------------------------------------------------ sharded optimizer state (data parallel, ZeRO-1)
    def enable_zero1(self, rank, world, adam8=False):
        """Data-parallel full fine-tune with the optimizer sharded over the ranks (the reference has no data parallelism to cite:
        full_finetuning_example.json names the workload).  Rank r owns the r-th of `world` equal slices of EVERY gradient bucket: the exchange
        step becomes reduce-scatter (each rank receives the summed gradients of its slices only) -> AdamW over the owned slices (p, g, m, v of
        1 / world of the matrices: 72 GB -> 72 / world GB of optimizer traffic per step on SDXL) -> all-gather of the updated fp32 masters ->
        the bf16 operand refresh of all matrices (20 GB).  Wire bytes equal the all-reduce's (which is a reduce-scatter + an all-gather);
        the moments exist for the owned slices only (2 x 10.3 GB -> 2 x 10.3 / world GB).  Every master element is computed by exactly
        one rank and copied to the others: replicas are bit-identical by construction.  The vector region (biases, norm affine parameters:
        0.1 % of the arena) stays replicated: all-reduce + the same AdamW everywhere."""
        assert self.params is not None and all((o1 - o0) % (4 * world) == 0 for o0, o1 in self.buckets), (world, self.buckets[:3])
        self.z_rank, self.z_world = rank, world
        self.z_chunk = [(o1 - o0) // world for o0, o1 in self.buckets]
        self.z_soff = [0]
        for c in self.z_chunk:
            self.z_soff.append(self.z_soff[-1] + c)
        ns, nm, dev = self.z_soff[-1], self.n_mat, self.rt.device
        if adam8:
            # AdamW8bit with the optimizer sharded: the owned slices' moments as byte codes + one fp32 absmax pair per 2048 consecutive elements of a slice
            # (sdlt_adamw8_flat: bitsandbytes' own flat partition, counted from the slice's start); the vector region keeps fp32 moments as in enable_8bit
            self.m_sh, self.v_sh = torch.zeros(ns, dtype=torch.uint8, device=dev), torch.zeros(ns, dtype=torch.uint8, device=dev)
            self.q8_sh = ([torch.zeros(2 * ((c + 2047) // 2048), dtype=F32, device=dev) for c in self.z_chunk], self.rt.ops.q8_tables(dev))
        else:
            self.m_sh, self.v_sh = torch.zeros(ns, dtype=F32, device=dev), torch.zeros(ns, dtype=F32, device=dev)
        self.m_vec, self.v_vec = self.m[nm:].clone(), self.v[nm:].clone()
        self.m = self.v = None           # the full-size moments are released

    def shard_range(self, b):
        """[s0, s1) of the arena: this rank's slice of bucket b."""
        o0, c = self.buckets[b][0], self.z_chunk[b]
        return o0 + self.z_rank * c, o0 + (self.z_rank + 1) * c

    def opt_state(self):
        if self.q8 is not None:
            return [self.q8[0], self.q8[1], self.q8[2], self.m_vec, self.v_vec]
        if self.m is None and self.q8_sh is not None:
            return [self.m_sh, self.v_sh, self.m_vec, self.v_vec] + list(self.q8_sh[0])
        return [self.m, self.v] if self.m is not None else [self.m_sh, self.v_sh, self.m_vec, self.v_vec]

    def adamw_shard_step(self, hyper):
        """AdamW over the owned slices (their summed gradients sit in `grads` after the in-place reduce-scatter) and over the replicated
        vector region; the masters of the other ranks' slices are stale until the all-gather, the bf16 operands until `refresh`."""
        ops, nm = self.rt.ops, self.n_mat
        for b in range(len(self.buckets)):
            s0, s1 = self.shard_range(b)
            so = self.z_soff[b]
            if self.q8_sh is not None:
                ops.adamw8_flat(self.params[s0:s1], self.grads[s0:s1], self.m_sh[so:so + s1 - s0], self.v_sh[so:so + s1 - s0], self.q8_sh[0][b], self.q8_sh[1], hyper)
            else:
                ops.adamw_fused(self.params[s0:s1], self.grads[s0:s1], self.m_sh[so:so + s1 - s0], self.v_sh[so:so + s1 - s0], hyper, None)
        if self.nv:
            ops.adamw_fused(self.params[nm:], self.grads[nm:], self.m_vec, self.v_vec, hyper, None)

    def zero_vector_grads(self):
        """Once per backward: bias and norm-affine gradients are accumulated (fp32 atomics) by the kernels that produce them."""
        self.grads[self.n_mat:].zero_()

    refresh_shadows = refresh      # same optimizer-facing surface as unet.LoraArena (params / grads / m / v / n / refresh_shadows)

    # ------------------------------------------------------------------ weight-gradient plan (leaf backward)
    def _panel(self, key, rows, Mp):
        n = rows * Mp
        return self.rt.scratch(key, (n + 1) // 2).view(BF16)[:n].view(rows, Mp)

    def linear(self, went, bent, xs, dy, n_rows=None):
        """went: [N, K] weight entry; xs: list of inputs whose channel concatenation is the layer input ([M, K_i] each);
        dy [M, >=N] (only the first N columns are the layer's output gradient)."""
        if self.defer:
            if self._wplan is None:
                self._jobs.append(("lin", went, bent, list(xs), dy, None))
            return
        ops = self.rt.ops
        N, K = went["shape"]
        M = dy.shape[0]
        Mp = _pad64(M)
        # the bias gradient (column sums of dY) falls out of the transposing pass over dY
        dyT = ops.wgrad_transpose(dy[:, :N] if dy.shape[1] != N else dy, self._panel("wg_dy", N, Mp),
                                  colsum_acc=self.view(bent, "grads") if bent is not None else None)
        gW = self.view(went, "grads")
        k0 = 0
        for x in xs:
            Ki = x.shape[1]
            xT = ops.wgrad_transpose(x, self._panel("wg_x", Ki, Mp))
            ops.gemm(dyT, xT, gW[:, k0:k0 + Ki] if len(xs) > 1 else gW)
            k0 += Ki
        assert k0 == K, (went["name"], k0, K)

    def conv3x3(self, went, bent, x, dy, *, B, H, W, Cin, stride, ups):
        """went: [Cout, 9*Cin] (tap-major); x NHWC [B*H*W, >=Cin]; dy [M, Cout_p]."""
        if self.defer:
            if self._wplan is None:
                self._jobs.append(("conv", went, bent, [x], dy, dict(B=B, H=H, W=W, stride=stride, ups=ups, Cin=Cin)))
            return
        ops = self.rt.ops
        Cout = went["shape"][0]
        M = dy.shape[0]
        Mp = _pad64(M)
        dyT = ops.wgrad_transpose(dy[:, :Cout] if dy.shape[1] != Cout else dy, self._panel("wg_dy", Cout, Mp),
                                  colsum_acc=self.view(bent, "grads"))
        cols = ops.wgrad_im2col_t(x[:, :Cin] if x.shape[1] != Cin else x, self._panel("wg_x", 9 * Cin, Mp), B=B, H=H, W=W,
                                  stride=stride, ups=ups)
        ops.gemm(dyT, cols, self.view(went, "grads"))

    def affine(self, *, groupnorm, x1, x2, dy, stats, gamma, beta, gent, bent, B, HW, eps=0.0, silu=False):
        """d gamma / d beta of one norm layer (accumulated into the pre-zeroed vector region): deferred and batched like the
        weight gradients - all LayerNorms of the 1280-wide blocks are ONE launch."""
        dg, db = self.view(gent, "grads"), self.view(bent, "grads")
        if self.defer:
            if self._aplan is None:
                self._affine_jobs.append(dict(gn=bool(groupnorm), x1=x1, x2=x2, dy=dy, stats=stats, gamma=gamma, beta=beta, dgamma=dg, dbeta=db,
                                              B=B, HW=HW, eps=eps, silu=bool(silu)))
            return
        ops = self.rt.ops
        if groupnorm:
            ops.groupnorm_affine_grad(x1, x2, dy, stats, dg, db, B=B, HW=HW, gamma=gamma, beta=beta, eps=eps, silu=silu, accumulate=True)
        else:
            ops.layernorm_affine_grad(x1, dy, stats, dg, db, accumulate=True)

    def _build_aplan(self, jobs):
        from collections import OrderedDict
        groups = OrderedDict()
        for j in jobs:
            key = (j["gn"], tuple(j["x1"].shape), j["x1"].stride(), None if j["x2"] is None else (tuple(j["x2"].shape), j["x2"].stride()),
                   tuple(j["dy"].shape), j["dy"].stride(), j["B"], j["HW"], j["eps"], j["silu"])
            groups.setdefault(key, []).append(j)
        return [self.rt.ops.AffineGradBatch(m, self.rt.device, groupnorm=m[0]["gn"], B=m[0]["B"], HW=m[0]["HW"], eps=m[0]["eps"], silu=m[0]["silu"])
                for m in groups.values()]

    def bucket_of(self, off):
        for i, (o0, o1) in enumerate(self.buckets):
            if o0 <= off < o1:
                return i
        raise ValueError(off)

    def flush(self, bucket=None):
        """End of a backward pass: run the weight-gradient plan recorded during the first one.  bucket=None: everything (affine
        gradients first).  bucket=i: the norm-affine gradients with i == 0, then only the layers whose weights lie in arena bucket i
        - afterwards grads[buckets[i]] is final (the vector region - biases, norm affine - only after the LAST bucket)."""
        if not self.defer:
            return
        if bucket in (None, 0):
            if self._aplan is None:
                self._aplan = self._build_aplan(self._affine_jobs)
                self._affine_jobs = None
            for ab in self._aplan:
                ab.run()
        if self._wplan is None:
            self._wplan = self._build_wplan(self._jobs)
            self._jobs = None
        ops = self.rt.ops
        for bidx, panels, gemms in self._wplan:
            if bucket is not None and bidx != bucket:
                continue
            for pb in panels:
                pb.run()
            for (X, W, C_, batch, tile) in gemms:
                ops.gemm(X, W, C_, batch=batch, tile=tile)

    def _build_wplan(self, jobs):
        from collections import OrderedDict
        rt, ops, dev = self.rt, self.rt.ops, self.rt.device
        groups = OrderedDict()
        for kind, went, bent, xs, dy, geom in jobs:
            N = went["shape"][0]
            dyv = dy[:, :N] if dy.shape[1] != N else dy
            if kind == "conv":
                xs = [xs[0][:, :geom["Cin"]] if xs[0].shape[1] != geom["Cin"] else xs[0]]
            key = (kind, tuple(dyv.shape), dyv.stride(), tuple((tuple(x.shape), x.stride()) for x in xs), bent is not None,
                   tuple(sorted(geom.items())) if geom else None, went["shape"])
            groups.setdefault(key, []).append((went, bent, xs, dyv, geom))
        plan = []
        for key, members in groups.items():
            kind, n = key[0], len(members)
            went0, _, xs0, dy0, geom = members[0]
            N, K = went0["shape"]
            M = dy0.shape[0]
            Mp = _pad64(M)
            dyT = torch.empty(n, N, Mp, dtype=BF16, device=dev)
            panels = [ops.WgradPanelBatch([(m[3], dyT[i], self.view(m[1], "grads") if m[1] is not None else None) for i, m in enumerate(members)], dev)]
            gemms, k0 = [], 0
            for j, x0 in enumerate(xs0):
                rows = (9 if kind == "conv" else 1) * x0.shape[1]
                xT = torch.empty(n, rows, Mp, dtype=BF16, device=dev)
                conv = dict(B=geom["B"], H=geom["H"], W=geom["W"], stride=geom["stride"], ups=geom["ups"]) if kind == "conv" else None
                panels.append(ops.WgradPanelBatch([(m[2][j], xT[i], None) for i, m in enumerate(members)], dev, conv=conv))
                outs = []
                for m in members:
                    gW = self.view(m[0], "grads")
                    outs.append(gW[:, k0:k0 + rows] if len(xs0) > 1 else gW)
                k0 += rows
                t128 = ((N + 127) // 128) * ((rows + 127) // 128)
                if n >= 4:          # enough problems to fill the chip without split-K: one batched launch
                    items = [dict(X=dyT[i], W=xT[i], C=outs[i]) for i in range(n)]
                    gemms.append((dyT[0], xT[0], outs[0], ops.GemmBatch(items, dev), 1 if t128 >= 8 else 3))
                else:               # a few long-K problems (the 128x128-resolution convs): individual launches keep their split-K
                    gemms += [(dyT[i], xT[i], outs[i], None, 0) for i in range(n)]
            assert k0 == K, (went0["name"], k0, K)
            bidx = self.bucket_of(went0["off"])
            assert all(self.bucket_of(m[0]["off"]) == bidx for m in members), "a weight-gradient group spans two arena buckets"
            plan.append((bidx, panels, gemms))
        plan.sort(key=lambda t: t[0])          # bucket by bucket (stable: first-appearance order inside a bucket)
        return plan

    # ------------------------------------------------------------------ host side: checkpoint layouts
    def export(self, which="params"):
        """-> {diffusers parameter name: fp32 tensor in the PyTorch layout}."""
        out = {}
        for e in self.entries:
            t = self.view(e, which).detach().float().cpu()
            if e["kind"] == "conv3x3":
                co, k9 = t.shape
                t = t.reshape(co, 3, 3, k9 // 9).permute(0, 3, 1, 2).contiguous()
            elif e["kind"] == "conv1x1":
                t = t.reshape(*t.shape, 1, 1)
            out[e["name"]] = t
        return out

    def load(self, sd):
        for e in self.entries:
            t = sd[e["name"]].to(self.rt.device, F32)
            if e["kind"] == "conv3x3":
                t = t.permute(0, 2, 3, 1).reshape(e["shape"])
            self.view(e).copy_(t.reshape(e["shape"]))
        self.refresh()
