"""sha256 over the kernel sources and the plan code (csrc/*.hip, *.h, *.cpp, the package's *.py, include/*.h): tools/step_profile.py stores it in the step profile,
bench.py recomputes it - `roofline.traffic_sources_match` says whether the profile a bench line quotes was taken on the code that produced the line."""
import glob
import hashlib
import os


def kernel_sources_sha():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    pk = os.path.join(root, "sd-lora-trainer_amd")
    for f in sorted(glob.glob(os.path.join(pk, "csrc", "*.hip")) + glob.glob(os.path.join(pk, "csrc", "*.h")) + glob.glob(os.path.join(pk, "csrc", "*.cpp"))
                    + glob.glob(os.path.join(pk, "*.py")) + glob.glob(os.path.join(root, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
