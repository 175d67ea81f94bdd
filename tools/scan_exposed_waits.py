"""Scan compiler output (hipcc -save-temps .s files) for global loads that are followed, within a few instructions, by `s_waitcnt vmcnt(0)`: an exposed round trip.  Round 5 found
three patterns with it: a load under a condition merged with a zero default (hipcc copies the loaded registers at the join and waits there: five serial round trips in the wave-split-K
prologue, three in the tiled GEMM's), loads issued right before a barrier that the compiler then waits for, and one-deep software prefetch in the row loops of the norm kernels.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -c csrc/<f>.hip -o /tmp/isa/<f>.o -save-temps=obj ; python tools/scan_exposed_waits.py /tmp/isa/<f>-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import subprocess
import sys

for f in sys.argv[1:]:
    kern, out = None, {}
    for ln in open(f):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern, idx, last_load = m.group(1), 0, None
            continue
        if kern is None:
            continue
        t = ln.strip()
        if not t or t[0] in ";.":
            continue
        idx += 1
        op = t.split()[0]
        if op.startswith("global_load") or op.startswith("buffer_load"):
            last_load = (idx, op)
        if op == "s_waitcnt" and "vmcnt(0)" in t and last_load and idx - last_load[0] <= 12 and "lds" not in last_load[1]:
            out.setdefault(kern, []).append(idx)
        if op == "s_endpgm":
            kern = None
    for k, v in sorted(out.items(), key=lambda kv: -len(kv[1])):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120]
        print(f"{len(v):3d} load -> vmcnt(0) within 12 instructions, at instruction {v[:12]}  {name}")
