"""TEST INFRASTRUCTURE - CPU restatement of bitsandbytes' `AdamW8bit` for the parity tests; nothing in the product path may import this file.

What it restates: `bnb.optim.AdamW8bit(params, lr, weight_decay)` as the reference builds it (trainer/optimizer.py:19-21; `unet_optimizer_type: "AdamW8bit"` in
train_configs/full_finetuning_example.json:14), i.e. bitsandbytes 0.43.1 (requirements.txt:21) `Optimizer2State("adam", ..., optim_bits=8, block_wise=True,
percentile_clipping=100, min_8bit_size=4096)`:
  * code books: `functional.create_dynamic_map(signed=True)` for the first moment, `create_dynamic_map(signed=False)` for the second (optimizer.py: `name2qmap`);
  * the blockwise update `kOptimizerStatic8bit2StateBlockwise<T, ADAM, 2048, 8>` (csrc/kernels.cu): dequantise with the block's old absmax, update the moments
    in fp32, take the block's new absmax, step the parameter from the UNquantised moments (`p += step_size * m / (sqrt(v) + correction2 * eps)`, then
    `p *= 1 - lr * weight_decay`), requantise to the nearest code; a first-moment code whose sign differs from m's moves one step towards m.

A property of the algorithm as restated (worth knowing when reading a parity report): an element whose first moment is below ~2.7e-7 of its block's largest rounds to the code
of 0 and, when NEGATIVE, is moved one step further by the sign rule (the code of 0 counts as positive) - to -5.5e-7 x absmax, away from its value; its second moment has long been
rounded to 0 (that happens below 1.6e-7 of the block's largest v, i.e. gradients below 4e-4 of the block's largest).  Its next update is then ~lr x 5.5e-7 x absmax_m / (eps + ...):
negligible while 5.5e-7 x absmax_m stays below eps = 1e-8 (gradients up to ~1e-2, every real training run), large on synthetic gradients of order 1.

PARITY UNPINNED: bitsandbytes is a third-party dependency that is absent from /root/reference and from this image (`import bitsandbytes` fails), the reference
holds no golden vector of its optimizer state, so this restatement follows the published algorithm as read and is pinned by structure only (256 distinct sorted
codes, symmetry, the decade layout).  One deliberate difference, shared with the HIP kernel it checks (include/sdlt_kernels.h: sdlt_adamw8_shadow_refresh): a
quantisation block is 32 rows x 64 columns of the weight's [N, K] view (2048 elements, bnb's block size) instead of 2048 consecutive elements of the flattened tensor.
(sdlt_adamw8_flat, the sharded optimizer's slices, uses consecutive elements: checked with the range viewed as a [n / 64, 64] matrix, whose 32 x 64 blocks are exactly that.)
"""
import math

import torch

F32 = torch.float32


def create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8):
    """bitsandbytes/functional.py create_dynamic_map (0.43.1), same arithmetic: fp32 linspace and means, python-float scale, one rounding to fp32, sorted."""
    data = []
    non_sign_bits = total_bits - 1                      # (the published code subtracts 1 whether signed or not)
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    for i in range(max_exponent_bits):
        fraction_items = int(2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        boundaries = torch.linspace(0.1, 1, fraction_items)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    if additional_items > 0:
        boundaries = torch.linspace(0.1, 1, additional_items + 1)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    data.append(0)
    data.append(1.0)
    assert len(data) == 2 ** total_bits
    data.sort()
    return torch.tensor(data, dtype=F32)


def nearest_code(x, code):
    """Index of the code nearest to x (fp32 midpoints; x on a midpoint takes the lower code)."""
    mids = (code[:-1] + code[1:]) * 0.5
    return torch.bucketize(x.contiguous(), mids, right=False)


BR, BC = 32, 64          # one quantisation block: 32 rows x 64 columns = 2048 elements


def _blocks(x, rows, cols):
    """[rows, cols] -> [nbr, nbc, BR, BC] (zero padded)."""
    Rp, Cp = -(-rows // BR) * BR, -(-cols // BC) * BC
    y = torch.zeros(Rp, Cp, dtype=x.dtype)
    y[:rows, :cols] = x
    return y.view(Rp // BR, BR, Cp // BC, BC).permute(0, 2, 1, 3).contiguous()


def _unblocks(y, rows, cols):
    nbr, nbc = y.shape[:2]
    return y.permute(0, 2, 1, 3).reshape(nbr * BR, nbc * BC)[:rows, :cols].contiguous()


class Adam8State:
    """Optimizer state of one matrix parameter [rows, cols]: byte codes and per-block absmax (zero before the first step, like bnb's init_state)."""

    def __init__(self, rows, cols):
        self.rows, self.cols = rows, cols
        self.m8 = torch.zeros(rows, cols, dtype=torch.uint8)
        self.v8 = torch.zeros(rows, cols, dtype=torch.uint8)
        nbr, nbc = -(-rows // BR), -(-cols // BC)
        self.am = torch.zeros(nbr, nbc, dtype=F32)
        self.av = torch.zeros(nbr, nbc, dtype=F32)

    def tile_absmax(self):
        """The kernel's layout: one row {m lo, m hi, v lo, v hi} per 64 x 64 tile, tiles row-major."""
        nbr, nbc = self.am.shape
        tr = -(-self.rows // 64)
        out = torch.zeros(tr, nbc, 4, dtype=F32)
        for h in range(2):
            sel = torch.arange(h, nbr, 2)
            out[: len(sel), :, h] = self.am[sel]
            out[: len(sel), :, 2 + h] = self.av[sel]
        return out.reshape(-1, 4)

    def set_tile_absmax(self, t):
        nbr, nbc = self.am.shape
        t = t.reshape(-1, nbc, 4)
        for h in range(2):
            sel = torch.arange(h, nbr, 2)
            self.am[sel] = t[: len(sel), :, h]
            self.av[sel] = t[: len(sel), :, 2 + h]


_MAPS = {}


def maps():
    if not _MAPS:
        _MAPS[True], _MAPS[False] = create_dynamic_map(True), create_dynamic_map(False)
    return _MAPS[True], _MAPS[False]


def adamw8_step(p, g, st, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1, grad_scale=1.0):
    """One AdamW8bit step on a matrix: returns the new fp32 parameter, updates `st` in place.  All arithmetic in fp32 like the kernel it restates."""
    rows, cols = st.rows, st.cols
    q1, q2 = maps()
    f = lambda x: torch.tensor(x, dtype=F32)  # noqa: E731
    valid = _blocks(torch.ones(rows, cols, dtype=torch.bool), rows, cols)
    gi = _blocks(g.to(F32), rows, cols) * f(grad_scale)
    m = q1[_blocks(st.m8, rows, cols).long()] * st.am[:, :, None, None]
    v = q2[_blocks(st.v8, rows, cols).long()] * st.av[:, :, None, None]
    m = f(beta1) * m + (1 - f(beta1)) * gi
    v = f(beta2) * v + (1 - f(beta2)) * gi * gi
    m = torch.where(valid, m, torch.zeros_like(m))
    v = torch.where(valid, v, torch.zeros_like(v))
    am, av = m.abs().amax(dim=(2, 3)), v.amax(dim=(2, 3))
    c1, c2 = 1.0 - beta1 ** step, math.sqrt(1.0 - beta2 ** step)
    step_size = f(-lr * c2 / c1)
    pn = _blocks(p.to(F32), rows, cols) + step_size * (m / (v.sqrt() + f(c2 * eps)))
    if weight_decay > 0.0:
        pn = pn * f(1.0 - lr * weight_decay)
    one = torch.ones_like(am)
    xm = torch.where(am[:, :, None, None] > 0, m / torch.where(am > 0, am, one)[:, :, None, None], torch.zeros_like(m))
    xv = torch.where(av[:, :, None, None] > 0, v / torch.where(av > 0, av, one)[:, :, None, None], torch.zeros_like(v))
    k1 = nearest_code(xm, q1)
    flip = (q1[k1] < 0) != (m < 0)                       # signbit(code) != signbit(m): the code of 0 counts as positive
    k1 = torch.where(flip, k1 + torch.where(m > 0, 1, -1), k1)
    k2 = nearest_code(xv, q2)
    st.m8 = _unblocks(k1.to(torch.uint8), rows, cols)
    st.v8 = _unblocks(k2.to(torch.uint8), rows, cols)
    st.am, st.av = am, av
    return _unblocks(pn, rows, cols)


def moments(st):
    """The fp32 moments the state encodes."""
    q1, q2 = maps()
    m = q1[_blocks(st.m8, st.rows, st.cols).long()] * st.am[:, :, None, None]
    v = q2[_blocks(st.v8, st.rows, st.cols).long()] * st.av[:, :, None, None]
    return _unblocks(m, st.rows, st.cols), _unblocks(v, st.rows, st.cols)
