"""Is the split GEMM power-bound?  Runs the persistent split-f16 GEMM (65536 x 1024 x 1024) back to back for a few seconds on (a) random and (b) all-zero
operands while a sampler thread reads the board power and the shader clock (rocm-smi / amdsmi / hwmon, whichever answers), then prints the mean
power, clock and TFLOP/s of each phase.  usage: gemm_power_sample.py [seconds per phase]"""
import sys, time, threading, subprocess, glob, re
import torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0

def read_hwmon():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try: out["power_w"] = int(open(f).read()) / 1e6
        except Exception: pass
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"):
        try: out["sclk_mhz"] = int(open(f).read()) / 1e6
        except Exception: pass
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            m = re.search(r"(\d+)Mhz \*", open(f).read())
            if m: out["dpm_sclk_mhz"] = float(m.group(1))
        except Exception: pass
    return out

def read_smi():
    out = {}
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        import json
        d = json.loads(t)
        for card, v in d.items():
            for k, val in v.items():
                kl = k.lower()
                if "power" in kl and "w" in kl:
                    try: out["smi_power_w"] = float(val)
                    except Exception: pass
                if "sclk" in kl:
                    m = re.search(r"(\d+)\s*mhz", str(val).lower())
                    if m: out["smi_sclk_mhz"] = float(m.group(1))
            break
    except Exception as e:
        out["smi_error"] = str(e)[:60]
    return out

samples, stop = [], False
def sampler():
    while not stop:
        s = read_hwmon(); s.update(read_smi()); s["t"] = time.time(); samples.append(s)
        time.sleep(0.2)

print("probe:", read_hwmon(), read_smi(), flush=True)
M, N, K = 65536, 1024, 1024
b = torch.randn(N, device="cuda")
for name, fill in (("random", True), ("zeros", False), ("random again", True)):
    if fill:
        a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    else:
        a = torch.zeros(M, K, device="cuda"); w = torch.zeros(N, K, device="cuda")
    fn = lambda: ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(50): fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    keys = sorted({k for s in samples for k in s if k not in ("t", "smi_error")})
    mean = {k: sum(s[k] for s in samples if k in s) / max(1, sum(1 for s in samples if k in s)) for k in keys}
    mx = {k: max((s[k] for s in samples if k in s), default=0) for k in keys}
    print(f"{name}: {us:.1f} us per launch = {6.0 * M * N * K / us / 1e6:.0f} TFLOP/s executed; {len(samples)} samples; mean {mean}; max {mx}", flush=True)
