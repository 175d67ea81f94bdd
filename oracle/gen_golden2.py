"""Round-2 fixtures generated from the REFERENCE's own code (read-only tree at /root/reference), like gen_golden.py:

  tests/golden/prompts.json      trainer/inference.py `prepare_prompt_for_lora`, trainer/utils/utils.py `fix_prompt`,
                                 `replace_in_string` on a set of prompts / concept modes / run names
  tests/golden/lr_schedule.json  SURVEY 8c (viii): the learning rates main.py:265-291 writes into `param_groups[0]['lr']`.  That code
                                 is inline in `train()`, so the statements of exactly those lines are cut out of main.py's AST HERE, at
                                 generation time, and executed against duck-typed optimizers - the table is the reference's own code's
                                 output, none of its source is stored
  tests/golden/token_init.pt     SURVEY 8c (x): `TokenEmbeddingsHandler.initialize_new_tokens` (embedding_handler.py:157-223) on a random-
                                 init transformers CLIPTextModel + CLIPTokenizer over a generated vocabulary: train ids, the std target,
                                 statistics of the initialised rows, the no-update index

Run:  python oracle/gen_golden2.py            (needs /root/reference; the GPU box never runs this)
"""
import ast
import json
import os
import sys
import tempfile
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gen_golden import OUT, REF, _install_stubs  # noqa: E402


def gen_prompts():
    import trainer.inference as rinf
    from trainer.utils.utils import fix_prompt, replace_in_string
    cases = []
    prompts = ["a photo of <concept> on the beach", "<concept>", "a painting in the style of <concept>, vivid", "My_Run riding a bike", "<my_run> and <concepts> hat",
               "a cat , sitting .on a mat,, outside", "portrait of TOK, studio light", "in the style of my_run , a castle", ""]
    for mode in ("object", "face", "style"):
        for name, trig in (("My_Run", "TOK"), ("banny", "TOK, "), ("my_style", "in the style of TOK, ")):
            with tempfile.TemporaryDirectory() as d:
                json.dump({"TOK": "<s0><s1><s2>"}, open(os.path.join(d, "special_params.json"), "w"))
                json.dump({"name": name, "concept_mode": mode, "training_attributes": {"trigger_text": trig}}, open(os.path.join(d, "training_args.json"), "w"))
                for p in prompts:
                    for interp in (False, True):
                        cases.append(dict(prompt=p, mode=mode, name=name, trigger_text=trig, interpolation=interp,
                                          out=rinf.prepare_prompt_for_lora(p, d, interpolation=interp, verbose=False)))
    fixes = [dict(inp=p, out=fix_prompt(p)) for p in ["a  b ,c,,d .e", " x , y ", "", "no change", "a.b.c , ,d"]]
    repl = [dict(s=s, r=r, out=replace_in_string(s, r)) for s, r in [("Foo foo FOO bar", {"foo": "x"}), ("<concept> and <Concept>", {"<concept>": "TOK"}),
                                                                       ("aXbXc", {"x": "yy", "b": "q"})]]
    json.dump(dict(prepare=cases, fix=fixes, replace=repl, negative_prompt=_negative_prompt()), open(os.path.join(OUT, "prompts.json"), "w"), indent=1)
    print("prompts.json:", len(cases), "prepare cases")


def _negative_prompt():
    """The string literal assigned to `negative_prompt` inside render_images (inference.py:362)."""
    tree = ast.parse(open(os.path.join(REF, "trainer", "inference.py")).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "negative_prompt" for t in node.targets) and isinstance(node.value, ast.Constant):
            return node.value.value
    raise RuntimeError("negative_prompt literal not found")


def gen_lr_schedule():
    """Executes main.py's own schedule statements (the `for step, batch` body up to `if not config.aspect_ratio_bucketing`)."""
    src = open(os.path.join(REF, "main.py")).read()
    tree = ast.parse(src)
    train_fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "train")
    inner = None
    for node in ast.walk(train_fn):
        if isinstance(node, ast.For) and isinstance(node.target, ast.Tuple) and [getattr(e, "id", None) for e in node.target.elts] == ["step", "batch"]:
            inner = node
    assert inner is not None
    body = []
    for st in inner.body:
        if isinstance(st, ast.If) and "aspect_ratio_bucketing" in ast.unparse(st.test):
            break
        if isinstance(st, ast.Expr) and "progress_bar" in ast.unparse(st):
            continue
        body.append(st)
    base = next(n for n in ast.walk(train_fn) if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == "base_unet_lr")
    base_if = next(n for n in train_fn.body if isinstance(n, ast.If) and "is_lora" in ast.unparse(n.test) and "base_unet_lr" in ast.unparse(n))
    code_base = compile(ast.Module([base, base_if], []), "main.py:236-240", "exec")
    code = compile(ast.Module(body, []), "main.py:265-291", "exec")

    class _Opt:
        def __init__(self):
            self.param_groups = [{"lr": -1.0}]

    tables = []
    variants = [dict(), dict(disable_ti=True), dict(is_lora=False, disable_ti=True), dict(ti_optimizer="prodigy"), dict(text_lora=True, txt_encoders_lr_warmup_steps=7),
                dict(freeze_unet_before_completion_f=0.25, freeze_ti_after_completion_f=0.5, unet_lr=1e-3, ti_lr=3e-3, unet_lr_warmup_steps=11)]
    for v in variants:
        cfg = types.SimpleNamespace(is_lora=v.get("is_lora", True), disable_ti=v.get("disable_ti", False), ti_optimizer=v.get("ti_optimizer", "adamw"),
                                    ti_lr=v.get("ti_lr", 1e-3), freeze_ti_after_completion_f=v.get("freeze_ti_after_completion_f", 0.7),
                                    text_encoder_lora_lr=1e-5, txt_encoders_lr_warmup_steps=v.get("txt_encoders_lr_warmup_steps", 200),
                                    unet_lr=v.get("unet_lr", 3e-4), unet_lr_warmup_steps=v.get("unet_lr_warmup_steps", 30),
                                    freeze_unet_before_completion_f=v.get("freeze_unet_before_completion_f", 0.0), num_train_epochs=5)
        env = dict(config=cfg)
        exec(code_base, env)
        optimizers = {"textual_inversion": None if cfg.disable_ti else _Opt(), "text_encoders": _Opt() if v.get("text_lora") else None, "unet": _Opt()}
        dl = list(range(6))
        rows, gs = [], 0
        for epoch in range(cfg.num_train_epochs):
            for step in range(len(dl)):
                env.update(optimizers=optimizers, epoch=epoch, step=step, train_dataloader=dl, global_step=gs)
                exec(code, env)
                rows.append([epoch, step, gs, env["completion_f"]] + [None if o is None else o.param_groups[0]["lr"] for o in optimizers.values()])
                gs += 1
        tables.append(dict(config={k: getattr(cfg, k) for k in vars(cfg)}, text_lora=bool(v.get("text_lora")), steps_per_epoch=len(dl), base_unet_lr=env["base_unet_lr"],
                           columns=["epoch", "step", "global_step", "completion_f", "lr_ti", "lr_text_encoders", "lr_unet"], rows=rows))
    json.dump(tables, open(os.path.join(OUT, "lr_schedule.json"), "w"))
    print("lr_schedule.json:", len(tables), "tables of", len(tables[0]["rows"]), "rows")


def gen_token_init():
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
    import trainer.embedding_handler as reh
    from tests.test_tokenizer_cpu import _train_bpe
    vocab, merges = _train_bpe(200)
    out = []
    for seed in (0, 7):
        toks = [CLIPTokenizer(vocab=dict(vocab), merges=list(merges)) for _ in range(2)]
        torch.manual_seed(100 + seed)
        encs = [CLIPTextModel(CLIPTextConfig(vocab_size=len(vocab), hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2)),
                CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=len(vocab), hidden_size=96, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, projection_dim=32))]
        for e in encs:         # make the pretrained rows non-uniform in scale so that the std target is not trivially the init std
            w = e.get_input_embeddings().weight.data
            w.mul_(torch.linspace(0.5, 2.0, w.shape[0])[:, None])
        pre = [e.get_input_embeddings().weight.data.clone() for e in encs]
        for e in encs:         # transformers 5.x flattened CLIPTextModel (no `.text_model`); the reference (4.38) reads `encoder.text_model.*`
            if not hasattr(e, "text_model"):
                object.__setattr__(e, "text_model", e)
        h = reh.TokenEmbeddingsHandler(encs, toks)
        h.initialize_new_tokens(inserting_toks=["<s0>", "<s1>", "<s2>"], starting_toks=None, seed=seed)
        rec = dict(seed=seed, vocab_size=len(vocab), train_ids=list(h.train_ids), pretrained=pre)
        for i, e in enumerate(encs):
            w = e.get_input_embeddings().weight.data
            rec[f"table_{i}"] = w.clone()                    # resized table after initialisation (the new rows are the last 3)
            rec[f"std_token_embedding_{i}"] = h.embeddings_settings[f"std_token_embedding_{i}"].clone()
            rec[f"index_no_updates_{i}"] = h.embeddings_settings[f"index_no_updates_{i}"].clone()
            rec[f"row_std_mean_{i}"] = w[h.train_ids].std(dim=1).mean()
        out.append(rec)
    torch.save(out, os.path.join(OUT, "token_init.pt"))
    print("token_init.pt: train ids", out[0]["train_ids"], "std targets", [float(out[0][f'std_token_embedding_{i}']) for i in range(2)],
          "row std means", [float(out[0][f'row_std_mean_{i}']) for i in range(2)])


if __name__ == "__main__":
    _install_stubs()
    gen_prompts()
    gen_lr_schedule()
    gen_token_init()
