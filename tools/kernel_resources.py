"""Registers / scratch / static LDS of every kernel variant of the given csrc files, from the compiler's own metadata (hipcc -S --cuda-device-only): the occupancy record behind the
DESIGN sections (waves per SIMD allowed by registers = min(8, floor(512 / alloc)), alloc = VGPR + AGPR rounded up to 8; dynamic LDS is set at launch and listed in the sources).
usage: python tools/kernel_resources.py wsk attn32 norm > profiles/r05_kernel_resources.txt"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "sd-lora-trainer_amd", "csrc")
for f in sys.argv[1:]:
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if f in ("attn", "attn32") else []
    out = f"/tmp/kr_{f}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", *extra, "-S", "--cuda-device-only",
                    os.path.join(csrc, f + ".hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    print(f"== {f}.hip")
    print(f"{'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS(static)':>12s} {'waves/SIMD':>11s}  kernel")
    for m in re.finditer(r"  - \.agpr_count:\s+(\d+)(.*?)\.vgpr_count:\s+(\d+)", txt, re.S):
        body = m.group(2)
        name = re.search(r"\.name:\s+(\S+)", body).group(1)
        sg = int(re.search(r"\.sgpr_count:\s+(\d+)", body).group(1))
        sc = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", body).group(1))
        ag, vg = int(m.group(1)), int(m.group(3))
        alloc = (vg + 7) // 8 * 8          # (.vgpr_count is the unified VGPR + AGPR budget on gfx950)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        print(f"{vg - ag:5d} {ag:5d} {sg:5d} {sc:8d} {lds:12d} {min(8, 512 // max(alloc, 1)):11d}  {dem[:150]}")
