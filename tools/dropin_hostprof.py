"""Host-side profile of the unchanged-trainer route (bench.py::drop_in_route): cProfile over a few steps, our frames first.
usage: python tools/dropin_hostprof.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
args = types.SimpleNamespace(warmup=3, steps=steps, no_conv_autotune=False)
bench.conv_autotune(args)
dev = torch.device("cuda", 0)
print("plain:", {k: v for k, v in bench.drop_in_route(args, dev).items() if k != "route"}, flush=True)
pr = cProfile.Profile()
pr.enable()
out = bench.drop_in_route(args, dev)
pr.disable()
print("profiled:", {k: v for k, v in out.items() if k != "route"}, flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(r"diffusion_finetuning_amd|harness|bench\.py", 70)
print("\n".join(line.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "") for line in s.getvalue().splitlines())[:14000], flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(40)
print(s.getvalue()[:8000], flush=True)
