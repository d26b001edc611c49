"""Dev tool (second half): per launch class, the time of a fused-GEMM launch with its main-loop loads, its MFMA work, or
both switched off (LORA_GEMM_DBG = 1 / 2 / 3: timing only — the results are wrong and the loss is garbage) — taken in the in-model
roofline pass of bench.py, so the launches sit between the same stock kernels as in a real step, next to the measured launch floor of
the same classes.  usage: python tools/skeleton_per_class.py > out.log      (one GPU, ≈ 3 min)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "full kernels", 1: "no main-loop loads", 2: "no MFMA work", 3: "neither loads nor MFMA"}
for mode in (0, 1, 2, 3):
    env = dict(os.environ, LORA_GEMM_DBG=str(mode))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-extra", "--no-cpu-baseline",
                          "--no-graph"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    if not lines:
        print(f"== LORA_GEMM_DBG={mode}: no bench line (exit code {out.returncode})")
        continue
    r = json.loads(lines[-1])["roofline"]
    sl = r["step_level"]
    print(f"== LORA_GEMM_DBG={mode} ({NAMES[mode]}): all 8(d) kernels {sl['kernel_ms_per_step']:.3f} ms per step, "
          f"launch floor {sl.get('launch_floor_ms', float('nan')):.3f} ms")
    for k, v in r["fused_gemm_classes"].items():
        print(f"   {k:72s} {v['launches_per_step']:6.1f} launches/step  {v['us_per_launch']:7.2f} us per launch   "
              f"floor {v.get('launch_floor_us_per_launch', float('nan')):5.2f}")
    sys.stdout.flush()
