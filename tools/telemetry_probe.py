"""Scratch: what the box's amdsmi / sysfs say about clocks and power, idle and under a replayed GEMM loop (feeds bench.GpuTelemetry)."""
import os, sys, time, json, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print("gpu_metrics:", {k: m[k] for k in m if any(s in k for s in ("gfxclk", "power", "temperature_hotspot", "throttle", "uclk", "socclk", "gfx_activity", "energy"))})
    for f in ("amdsmi_get_power_cap_info", "amdsmi_get_power_info"):
        try: print(f, getattr(amdsmi, f)(h))
        except Exception as e: print(f, "failed:", e)
    try: print("clock_info", amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
    except Exception as e: print("clock_info failed:", e)
except Exception as e:
    print("amdsmi unavailable:", e)
print("sysfs:", glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"), glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1*"))
t = bench.GpuTelemetry()
print("static", t.static)
t.start(); time.sleep(0.5); print("idle", json.dumps(t.stop()))
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
t.start()
for _ in range(400): a @ b
torch.cuda.synchronize()
print("busy (hipBLASLt 8192^3 x400)", json.dumps(t.stop()))
