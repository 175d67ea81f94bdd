"""Textual-inversion token bookkeeping, mirroring trainer/embedding_handler.py (TokenEmbeddingsHandler:13-62, 157-223,
401-456) on top of the engine's text encoders (clip.ClipTextEncoder) and TI state (ti.TiState).

The reference grows each tokenizer/embedding table by `<s0>..<s{n-1}>` and initialises the new rows as
randn * sigma_bar / std_cur (embedding_handler.py:209-213).  Here the tables are built with the n extra rows already in
place (the last rows), so `initialize_new_tokens` only draws those rows."""
import torch
from safetensors.torch import save_file

from .checkpoint import load_embeddings as _load_embeddings


class TokenEmbeddingsHandler:
    def __init__(self, ti_state, inserting_toks):
        self.ti, self.inserting_toks = ti_state, list(inserting_toks)
        self.encoders = ti_state.encoders
        enc = self.encoders[0]
        self.train_ids = list(range(enc.V - len(inserting_toks), enc.V))     # the new tokens are the last rows
        self.embeddings_settings = {}

    def initialize_new_tokens(self, seed=0):
        """rows = randn * sigma_bar / (mean per-row std of the draw), sigma_bar = mean per-row std of the RESIZED table - the
        reference takes it after `resize_token_embeddings`, i.e. over all rows incl. the not yet initialised new ones
        (embedding_handler.py:183, 195-197; 3 rows of 49,411).  The draw itself is `torch.randn` after `seed_everything(seed)`
        (:172, 210); the reference's global generator has by then also been advanced by transformers' resize initialisation, which
        depends on the transformers version, so the statistics are what is pinned (tests/golden/token_init.pt), not the bits.
        Afterwards the regulariser's target statistics are taken over the whole table, as `ConditioningRegularizer` is built after
        the initialisation (main.py:92-105, loss.py:190-193)."""
        g = torch.Generator().manual_seed(seed)            # seed_everything(seed) in the reference (:172)
        rows = []
        n = len(self.train_ids)
        for idx, e in enumerate(self.encoders):
            std_token_embedding = e.table.float().std(dim=1).mean()
            self.embeddings_settings[f"std_token_embedding_{idx}"] = std_token_embedding
            init = torch.randn(n, e.D, generator=g)
            init = init * float(std_token_embedding) / init.std(dim=1).mean()
            rows.append(init)
            inu = torch.ones(e.V, dtype=torch.bool)
            inu[self.train_ids] = False
            self.embeddings_settings[f"index_no_updates_{idx}"] = inu
        self.ti.load_rows(rows)
        self.ti.set_reference_stats(whole_table=True)
        return rows

    def get_trainable_embeddings(self):
        """{'txt_encoder_i': rows [n, D]} like embedding_handler.py:37-62 (fp32 master rows)."""
        return {f"txt_encoder_{i}": r for i, r in enumerate(self.ti.rows)}

    def save_embeddings(self, file_path, txt_encoder_keys=("clip_l", "clip_g")):
        """The trained rows in the dtype of the encoder's embedding table, as the reference saves `token_embedding.weight.data[train_ids]`
        (embedding_handler.py:401-422: the table lives in `weight_type`, bf16 by default) - not the fp32 master rows."""
        save_file({txt_encoder_keys[i]: r.detach().to(self.encoders[i].table.dtype).cpu().contiguous() for i, r in enumerate(self.ti.rows)}, file_path)

    def load_embeddings(self, file_path, txt_encoder_keys=("clip_l", "clip_g")):
        self.ti.load_rows(_load_embeddings(file_path, txt_encoder_keys))
