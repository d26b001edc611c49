"""Reduces rocprofv3 --pmc passes of tools/gemm_bench.py to one line per launch class (kernel tile, grid): averages per
dispatch of every counter collected.  usage: python tools/pmc_gemm.py <counter_collection.csv> [...]"""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "lora_gemm_kernel" not in r["Kernel_Name"]:
            continue
        m = re.search(r"I(DF16_|DF16b|f)Li(\d+)ELi(\d+)ELb(\d)ELi(\d)", r["Kernel_Name"])
        key = (f"{m.group(2)}x{m.group(3)} s{m.group(5)}" if m else r["Kernel_Name"][:30], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        a = agg[key][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
print("tile / blocks".ljust(22) + " ".join(n.replace("SQ_", "").replace("_CYCLES", "").replace("_sum", "")[:14].rjust(15) for n in names))
for k in sorted(agg, key=lambda k: (k[0], k[1])):
    v = agg[k]
    print(f"{k[0]:12s} {k[1]:8d} " + " ".join(f"{v[n][1] / v[n][0]:15.0f}" if n in v else " " * 15 for n in names))
