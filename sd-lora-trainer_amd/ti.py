"""Textual-inversion state: the trainable token-embedding rows of every text encoder, their AdamW moments and the
token-std regulariser.

reference: trainer/embedding_handler.py:13-62 (TokenEmbeddingsHandler: which rows are trainable),
trainer/optimizer.py:107-155 (AdamW over the WHOLE embedding tables, weight decay ti_weight_decay = 0),
main.py:368-371 (gradient rows [:-n_tokens] zeroed), trainer/loss.py:196-233, 254-297 (std regulariser).
With zeroed gradients and zero weight decay every non-TI row is an exact no-op of AdamW, so only the n_tokens rows are
stored and stepped here (pinned by tests/golden/adamw.pt: rows-only == masked full-table update).
"""
import torch

from .unet import F32


class TiState:
    def __init__(self, rt, encoders, n_tok, std_loss_w=0.01):
        self.rt, self.encoders, self.n_tok = rt, encoders, n_tok
        sizes = [n_tok * e.D for e in encoders]
        self.n = sum(sizes)
        self.params, self.grads = rt.zeros(self.n, dtype=F32), rt.zeros(self.n, dtype=F32)
        self.m, self.v = rt.zeros(self.n, dtype=F32), rt.zeros(self.n, dtype=F32)
        self.rows, self.grad_rows, self.stats = [], [], []
        sh, off = [], 0
        for e, sz in zip(encoders, sizes):
            r = self.params[off:off + sz].view(n_tok, e.D)
            r.copy_(e.table[e.V - n_tok:].float())
            self.rows.append(r)
            self.grad_rows.append(self.grads[off:off + sz].view(n_tok, e.D))
            sh.append((off, n_tok, e.D, e.D, e.table[e.V - n_tok:], None))
            self._pretrained = getattr(self, "_pretrained", []) + [(e, e.V - n_tok)]
            off += sz
        self.set_reference_stats(whole_table=False)
        self._plan = rt.ops.ShadowPlan(sh, rt.device)
        self.std_loss_w = std_loss_w
        self.reg_loss, self.cov_loss = rt.zeros(1, dtype=F32), rt.zeros(1, dtype=F32)
        self.hyper = rt.zeros(16, dtype=F32)

    def set_reference_stats(self, whole_table):
        """DistributionLoss statistics (loss.py:263-265): mean of the per-row std and its normalised variance, over the pretrained
        rows - or over the WHOLE table incl. the freshly initialised new rows, which is what the reference's regulariser sees
        (it is constructed after `initialize_new_tokens`, main.py:92-105; TokenEmbeddingsHandler.initialize_new_tokens calls this)."""
        self.stats, self._target_cov = [], None
        pre = []
        for e, nv in self._pretrained:
            stds = (e.table if whole_table else e.table[:nv]).float().std(-1)
            self.stats.append((float(stds.mean()), float(stds.std() ** 2 / stds.mean())))
            pre.append((e, e.V if whole_table else nv))
        self._cov_rows = pre

    def refresh_tables(self):
        """fp32 master rows -> the bf16 embedding tables the encoders gather from."""
        self._plan.run(self.params)

    def load_rows(self, rows_per_encoder):
        for r, src in zip(self.rows, rows_per_encoder):
            r.copy_(src.to(self.rt.device, F32))
        self.refresh_tables()

    def add_regulariser(self, std_loss_w=None):
        """loss += std_loss_w * mean_enc( mean_tok( (sigma_bar - std(e_tok))^2 / v ) )   (loss.py:223-231); gradient
        added to the row gradients, value accumulated in self.reg_loss.  std_loss_w defaults to the step's 0.01; the
        token warm-up passes 0.5 (embedding_handler.py:384)."""
        self.reg_loss.zero_()
        w = (self.std_loss_w if std_loss_w is None else std_loss_w) / len(self.encoders)
        for r, g, (tm, tv) in zip(self.rows, self.grad_rows, self.stats):
            self.rt.ops.ti_std_reg(r, g, self.reg_loss, target_mean=tm, target_var=tv, weight=w)

    def add_covariance(self, weight):
        """tok_cov_reg_w term (trainer/loss.py:213-221, 275-289; weight 0 by default): loss += weight * mean_enc( || Cov(pretrained
        table) - Cov(rows) ||_F / D^2 ), gradient added to the row gradients.  [n, D] x [D, D] arithmetic on the trainable rows:
        torch ops (the target covariance of each table is built once, lazily)."""
        if getattr(self, "_target_cov", None) is None:
            self._target_cov = []
            for e, nv in self._cov_rows:
                t = e.table[:nv].float()
                adj = t - t.mean(0)
                self._target_cov.append(adj.T @ adj / (nv - 1))
        self.cov_loss.zero_()
        w = weight / len(self.encoders)
        for r, g, T in zip(self.rows, self.grad_rows, self._target_cov):
            n, D = r.shape
            A = r - r.mean(0)
            diff = T - A.T @ A / (n - 1)
            nrm = torch.linalg.norm(diff)
            self.cov_loss += w * nrm / (D * D)
            G = -diff / (nrm * D * D)                    # d loss / d Cov(rows)
            dA = A @ (G + G.T) / (n - 1)
            g.add_(w * (dA - dA.mean(0)))
