"""ClipBpeTokenizer (sd-lora-trainer_amd/tokenizer.py) against the installed `transformers.CLIPTokenizer` on a generated
byte-level BPE vocabulary (no vocabulary files exist offline): ids of plain / punctuated / accented / emoji prompts, padding
to 77 with truncation, and the trigger tokens `<s0><s1><s2>` appended as the reference does (embedding_handler.py:176-180)."""
import collections
import json
import os

import pytest

from sd_lora_trainer_amd.tokenizer import ClipBpeTokenizer, bytes_to_unicode

transformers = pytest.importorskip("transformers")

CORPUS = ("a photo of a cat sitting on the grass in front of a house . an oil painting of a woman , portrait , detailed face . "
          "the quick brown fox jumps over the lazy dog ! it's a dog's life , isn't it ? we're here , they've gone , i'm fine , you'll see , he'd know . "
          "photo of tok person wearing sunglasses at the beach 2023 , 35mm , f/1.8 , bokeh . in the style of tok , vibrant colors , 4k , 8k . "
          "café naïve façade über straße 東京 🙂 ") * 3


def _train_bpe(n_merges):
    import regex as re
    from sd_lora_trainer_amd.tokenizer import PATTERN
    be = bytes_to_unicode()
    words = collections.Counter()
    for tok in PATTERN.findall(re.sub(r"\s+", " ", CORPUS).lower()):
        sym = [be[b] for b in tok.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _c = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        merges.append((a, b))
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] += c
        words = new
    alphabet = list(be.values())
    vocab = {s: i for i, s in enumerate(alphabet + [s + "</w>" for s in alphabet] + [a + b for a, b in merges])}
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    return vocab, merges


PROMPTS = ["a photo of a cat", "A Photo   of\ta CAT, sitting on the grass!!", "it's a dog's life, isn't it? we're here", "café naïve façade über straße",
           "東京 🙂 emoji and cjk", "35mm f/1.8 bokeh 2023 4k", "", "in the style of <s0><s1><s2>, vibrant colors", "<s0><s1><s2>", "photo of <s0> <s1>  <s2> person",
           " ".join(["very long prompt with many words"] * 20)]


def test_matches_transformers_clip_tokenizer(tmp_path):
    vocab, merges = _train_bpe(300)
    d = tmp_path / "tokenizer"
    os.makedirs(d)
    with open(d / "vocab.json", "w", encoding="utf-8") as fh:
        json.dump(vocab, fh, ensure_ascii=False)
    with open(d / "merges.txt", "w", encoding="utf-8") as fh:
        fh.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    mine = ClipBpeTokenizer.from_files(str(d))
    ref = transformers.CLIPTokenizer(vocab=vocab, merges=merges)
    assert len(mine) == len(ref)
    toks = ["<s0>", "<s1>", "<s2>"]
    assert mine.add_tokens(toks) == ref.add_tokens(toks) == 3
    assert mine.convert_tokens_to_ids(toks) == ref.convert_tokens_to_ids(toks) == [len(vocab), len(vocab) + 1, len(vocab) + 2]
    for p in PROMPTS:
        assert mine.encode(p) == ref.encode(p), p
        want = ref(p, padding="max_length", max_length=77, truncation=True)["input_ids"]
        assert mine(p)[0] == want, p
    rows = mine(PROMPTS)
    assert all(len(r) == 77 and r[0] == mine.bos_token_id for r in rows)
    assert rows[-1][-1] == mine.eos_token_id                       # truncated prompt still ends with eos
    # SDXL's tokenizer_2 pads with "!" (id 0)
    t2 = ClipBpeTokenizer(vocab, [f"{a} {b}" for a, b in merges], pad_token="!")
    assert t2("a cat")[0][-1] == vocab["!"] and t2.pad_token_id == vocab["!"]
