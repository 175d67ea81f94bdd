"""CLIP byte-level BPE tokenizer from a `vocab.json` + `merges.txt` pair (the files of `openai/clip-vit-large-patch14` that
diffusers ships inside every SD1.5 / SDXL checkpoint as `tokenizer/` and `tokenizer_2/`), with the trigger tokens `<s0>..`
appended the way the reference does it (/root/reference trainer/embedding_handler.py:157-223: `tokenizer.add_tokens`; new ids
are vocab_size .. vocab_size + n - 1 = the LAST rows of the grown embedding tables).

This is the prompt -> ids half of `pipe.encode_prompt(prompt: str ...)` (trainer/inference.py:131-139) and of
`pipe.tokenizer.encode(caption)` (trainer/loss.py:32).  There are no vocabulary files offline, so the paths are configuration
(`pretrained_model["tokenizer_path"]`, `["tokenizer_2_path"]`: directories holding vocab.json / merges.txt).

The algorithm is CLIP's published one (OpenAI simple_tokenizer / Hugging Face CLIPTokenizer): NFC-normalise, collapse
whitespace, lower-case; split with the CLIP pattern; map each piece's UTF-8 bytes to the printable byte alphabet; merge
greedily by merge rank with `</w>` marking the last symbol of a word; look the symbols up in the vocabulary.  Checked
against the installed `transformers.CLIPTokenizer` on a generated vocabulary in tests/test_tokenizer_cpu.py.
"""
import json
import os
import unicodedata

import regex as re

PATTERN = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""")
MAX_LENGTH = 77


def bytes_to_unicode():
    """The reversible byte -> printable unicode character table of GPT-2 / CLIP byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class ClipBpeTokenizer:
    def __init__(self, vocab, merges, *, bos_token="<|startoftext|>", eos_token="<|endoftext|>", pad_token=None, max_length=MAX_LENGTH):
        """vocab: dict token -> id; merges: list of "a b" strings in rank order (without the `#version` header line).
        pad_token: `<|endoftext|>` for SD1.5 / SDXL tokenizer, "!" (id 0) for SDXL's tokenizer_2."""
        self.vocab = dict(vocab)
        self.ranks = {tuple(m.split()): i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.bos_token_id, self.eos_token_id = self.vocab[bos_token], self.vocab[eos_token]
        self.unk_token_id = self.eos_token_id
        self.pad_token_id = self.vocab[pad_token] if pad_token is not None else self.eos_token_id
        self.max_length = max_length
        self.added = {}                 # added token string -> id (matched verbatim before the BPE)
        self._cache = {}

    @classmethod
    def from_files(cls, path, **kw):
        """path: a directory with vocab.json + merges.txt (diffusers `tokenizer/` layout)."""
        with open(os.path.join(path, "vocab.json"), encoding="utf-8") as fh:
            vocab = json.load(fh)
        with open(os.path.join(path, "merges.txt"), encoding="utf-8") as fh:
            lines = fh.read().split("\n")
        merges = [l for l in lines if l and not l.startswith("#version")]
        if "pad_token" not in kw and os.path.exists(os.path.join(path, "special_tokens_map.json")):
            with open(os.path.join(path, "special_tokens_map.json"), encoding="utf-8") as fh:
                pad = json.load(fh).get("pad_token")
            pad = pad.get("content") if isinstance(pad, dict) else pad
            if pad in vocab:
                kw["pad_token"] = pad
        return cls(vocab, merges, **kw)

    def __len__(self):
        return len(self.vocab) + len(self.added)

    def add_tokens(self, tokens):
        """embedding_handler.py:176-180: the new tokens take the next free ids; returns how many were added."""
        n = 0
        for t in tokens:
            if t not in self.vocab and t not in self.added:
                self.added[t] = len(self)
                n += 1
        return n

    def convert_tokens_to_ids(self, tokens):
        one = isinstance(tokens, str)
        ids = [self.added.get(t, self.vocab.get(t, self.unk_token_id)) for t in ([tokens] if one else tokens)]
        return ids[0] if one else ids

    # ------------------------------------------------------------------ BPE
    def _bpe(self, token):
        if token in self._cache:
            return self._cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for i in range(len(word) - 1):
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (word[i], word[i + 1]), r
            if best is None:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self._cache[token] = word
        return word

    def _split_added(self, text):
        """-> list of (piece, is_added_token): added tokens are cut out verbatim, longest first."""
        if not self.added:
            return [(text, False)]
        pat = "|".join(re.escape(t) for t in sorted(self.added, key=len, reverse=True))
        out, pos = [], 0
        for m in re.finditer(pat, text):
            if m.start() > pos:
                out.append((text[pos:m.start()], False))
            out.append((m.group(0), True))
            pos = m.end()
        if pos < len(text):
            out.append((text[pos:], False))
        return out

    def tokenize_ids(self, text):
        """ids of the text WITHOUT bos / eos."""
        ids = []
        for piece, is_added in self._split_added(text):
            if is_added:
                ids.append(self.added[piece])
                continue
            piece = re.sub(r"\s+", " ", unicodedata.normalize("NFC", piece)).lower()
            for tok in PATTERN.findall(piece):
                if tok in ("<|startoftext|>", "<|endoftext|>"):
                    ids.append(self.vocab[tok])
                    continue
                sym = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
                ids.extend(self.vocab.get(s, self.unk_token_id) for s in self._bpe(sym))
        return ids

    def encode(self, text):
        """`tokenizer.encode(caption)` (trainer/loss.py:32): [bos] + ids + [eos], no padding, no truncation."""
        return [self.bos_token_id] + self.tokenize_ids(text) + [self.eos_token_id]

    def __call__(self, texts, max_length=None):
        """`tokenizer(prompt, padding="max_length", max_length=77, truncation=True)` of diffusers' encode_prompt: int64 rows
        [bos, ids..., eos, pad...] of length max_length (a longer prompt is cut and still ends with eos).  -> list of lists."""
        L = max_length or self.max_length
        rows = []
        for t in ([texts] if isinstance(texts, str) else texts):
            ids = self.tokenize_ids(t)[: L - 2]
            row = [self.bos_token_id] + ids + [self.eos_token_id]
            rows.append(row + [self.pad_token_id] * (L - len(row)))
        return rows
