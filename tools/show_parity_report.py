"""Summary of gpurun_out/parity_report.json (written by tests/test_real_topology_gpu.py): per case and oracle mode the first-step figures and the worst adapters."""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_report.json"))
for k, v in d.items():
    for o, r in v.items():
        if o == "displacement_after_steps":
            print(k, "LoRA displacement", r)
            continue
        print(k, o, "pred", round(r["pred_err"], 4), "cos", round(r["lora_cos"], 5), "rel", round(r["lora_rel"], 4), "median adapter rel", round(r["median_adapter_rel"], 4),
              "token rows (cos, rel)", [(round(a, 4), round(b, 4)) for a, b in r["token_rows"]], "loss rel", r.get("loss_rel"))
        for w in r["worst_adapters"][:3]:
            print("     ", w)
