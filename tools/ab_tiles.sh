for v in 1 -1 1 -1; do
LORA_FORCE_TILE=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$v.log 2> gpurun_out/ab_$v.err
python - <<P
import json
d=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
print("tile=$v", round(d["value"],2), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["avg_us"],2), round(d["hot_path"]["kernel_ms_per_step"],3), {k[17:28]:(v["launches_per_step"], round(v["avg_us"],1)) for k,v in d["hot_path"]["kernels"].items() if "gemm" in k})
P
done
