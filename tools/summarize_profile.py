"""Turns rocprofv3 CSV output of `bench.py` into the small summaries committed under profiles/.

  python tools/summarize_profile.py trace <kernel_trace.csv> <warmup_steps> <out.csv>
      per-kernel stats of the TIMED region only (from the (warmup+1)-th add_noise dispatch on), so one-time
      MIOpen solver searches during warm-up do not pollute the table.
  python tools/summarize_profile.py gaps <kernel_trace.csv> <warmup_steps>
      idle time between consecutive dispatches of the timed region (wall of the region against the sum of kernel durations,
      a histogram of the gaps and the kernels that the largest ones follow): what the replayed graph loses between nodes.
  python tools/summarize_profile.py pmc <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<mfma_counter_collection.csv>]
      average FETCH_SIZE / WRITE_SIZE (KB) per dispatch for the hot-path kernels, and the corrected traffic
      (2·FETCH_SIZE + WRITE_SIZE)·1024 bytes (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads ½ on gfx950); with a third
      pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) also the MFMA-busy share per kernel.
      The digest of csrc/ + the header the library was built from is stored under "_csrc_digest": bench.py refuses a
      traffic figure whose digest differs from the library it is running.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(lora_\w+kernel|attn_\w+kernel|ddpm_mse_kernel|add_noise_kernel|noise_prologue_kernel|reduce_partials_kernel|pack_factors\w*|grad_sqnorm_kernel|adamw_kernel)", name)
    if not m:
        return name[:80]
    t = re.search(r"I(DF16_|DF16b|f)(?:Li(\d+)E)?(?:Li(\d+)E)?(?:Lb(\d)E)?(?:Lb(\d)E)?", name)
    out = m.group(1) + ("<" + ",".join(x for x in t.groups() if x) + ">" if t else "")
    # the fused GEMM's remaining template arguments: ring depth, the GEGLU-gate form (0 none, 1 forward, 2 backward), split-K
    g = re.search(r"lora_gemm_kernelI(?:DF16_|DF16b|f)Li\d+ELi\d+ELb\dELi(\d+)ELi\d+ELi\d+ELi(\d+)E(?:Lb(\d)E)?(?:Li(\d+)E)?", name)
    if g:
        out = out[:-1] + f",s{g.group(1)},g{g.group(2)}" + (",splitk" if g.group(3) == "1" else "") + \
            (f",kp{g.group(4)}" if g.group(4) not in (None, "1") else "") + ">"
    return out


def _step_start(name):  # the first hot-path kernel of a train step: the noise prologue (device draw) or add_noise (host draw)
    return "add_noise_kernel" in name or "noise_prologue_kernel" in name


def trace(path, warmup, out):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [int(r["Start_Timestamp"]) for r in rows if _step_start(r["Kernel_Name"])]
    t0 = starts[warmup]
    steps = len(starts) - warmup
    agg = collections.defaultdict(list)
    for r in rows:
        if int(r["Start_Timestamp"]) >= t0:
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls_per_step", "avg_us", "min_us", "max_us", "ms_per_step", "pct_of_gpu_time"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, f"{len(v) / steps:.1f}", f"{sum(v) / len(v) / 1e3:.2f}", f"{min(v) / 1e3:.2f}", f"{max(v) / 1e3:.2f}",
                        f"{sum(v) / steps / 1e6:.3f}", f"{100 * sum(v) / total:.2f}"])
        w.writerow(["TOTAL (timed region, %d steps)" % steps, "", "", "", "", f"{total / steps / 1e6:.3f}", "100"])
    print(open(out).read())


def gaps(path, warmup):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [int(r["Start_Timestamp"]) for r in rows if _step_start(r["Kernel_Name"])]
    t0 = starts[warmup]
    steps = len(starts) - warmup
    reg = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows if int(r["Start_Timestamp"]) >= t0]
    # the region ends with the last dispatch of the last step; kernels may overlap (two queues): track the running end
    busy = 0
    idle = []
    end = reg[0][0]
    for (a, b, name), prev in zip(reg, [None] + reg[:-1]):
        if a > end:
            idle.append((a - end, prev[2] if prev else "", name))
            busy += b - a
        else:
            busy += max(0, b - end)
        end = max(end, b)
    wall = end - reg[0][0]
    tot_idle = sum(g for g, _, _ in idle)
    print(f"timed region: {steps} steps, wall {wall / steps / 1e6:.3f} ms/step, GPU busy {busy / steps / 1e6:.3f} ms/step, "
          f"idle between dispatches {tot_idle / steps / 1e6:.3f} ms/step ({100 * tot_idle / wall:.1f} %), "
          f"{len(reg) / steps:.0f} dispatches/step, {len(idle) / steps:.0f} gaps/step")
    edges = [0, 500, 1000, 2000, 4000, 8000, 16000, 10 ** 12]
    for lo, hi in zip(edges, edges[1:]):
        sel = [g for g, _, _ in idle if lo <= g < hi]
        print(f"  gaps {lo / 1e3:5.1f} - {hi / 1e3 if hi < 10 ** 11 else float('inf'):5.1f} us: {len(sel) / steps:7.1f} per step, {sum(sel) / steps / 1e6:.3f} ms/step")
    by = collections.defaultdict(lambda: [0, 0])
    for g, prev, nxt in idle:
        by[(prev[:60], nxt[:60])][0] += 1
        by[(prev[:60], nxt[:60])][1] += g
    print("  largest contributors (previous kernel -> next kernel):")
    for (prev, nxt), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f"    {t / steps / 1e3:8.1f} us/step  {n / steps:6.1f} x  {t / n / 1e3:6.2f} us  {prev}  ->  {nxt}")


def pmc(fetch, write, out, mfma=None):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffusion_finetuning_amd.build_native import _digest

    def load(path, counter):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and any(t in r["Kernel_Name"] for t in ("lora_", "ddpm", "attn_")):
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        return agg
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {"_csrc_digest": _digest()}
    for k in f:
        fk, wk = f[k][1] / f[k][0], w[k][1] / max(1, w[k][0])
        res[k] = {"dispatches": f[k][0], "FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk,
                  "traffic_bytes_per_launch": (2 * fk + wk) * 1024}
    if mfma:
        # MFMA-busy share: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed
        # over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back) -> busy / (GUI_ACTIVE/8 * 1024 SIMDs)
        busy, sqb, gui = (load(mfma, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
        for k in busy:
            n = busy[k][0]
            b, g = busy[k][1] / n, gui[k][1] / max(1, gui[k][0])
            res.setdefault(k, {"dispatches": n}).update({
                "SQ_VALU_MFMA_BUSY_CYCLES_avg": b, "SQ_BUSY_CYCLES_avg": sqb[k][1] / max(1, sqb[k][0]),
                "GRBM_GUI_ACTIVE_avg": g, "mfma_busy_frac": b / (g / 8 * 1024) if g else None})
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def sq(paths):
    """SQ wait / active counters per hot-path kernel from one or more --pmc passes of bench.py: sums per kernel class and the
    ratios that say what a wave's life goes to (SQ_* wave counters are in quad-cycles, MI355X_MICROARCH.md): WAIT_ANY / WAVE_CYCLES
    = share spent on s_waitcnt (operands in flight), WAIT_INST_ANY / WAVE_CYCLES = waiting for an issue slot.  Every pass collects
    SQ_WAVE_CYCLES itself: a counter is divided by the wave cycles of ITS pass."""
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    ratio = collections.defaultdict(dict)
    n = collections.defaultdict(int)
    for path in paths:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = set()
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if not re.search(r"lora_|attn_|ddpm_", name):
                continue
            k = short(name)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add((k, r.get("Dispatch_Id")))
        for k, _ in seen:
            n[k] += 1
        for k, v in agg.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            for c, x in v.items():
                sums[k][c] += x
                if wc and c != "SQ_WAVE_CYCLES":
                    ratio[k][c.replace("SQ_", "") + "/WAVE"] = round(x / wc, 3)
    print("# kernel class: dispatches (over the passes), ratios to the SQ_WAVE_CYCLES of the counter's own pass, counter sums")
    for k in sorted(sums, key=lambda k: -sums[k].get("SQ_WAVE_CYCLES", 0.0)):
        print(f"{k:52s} n={n[k]:5d} {ratio[k]} " + " ".join(f"{c}={int(x)}" for c, x in sorted(sums[k].items())))


def shapes(path, warmup):
    """In-model time of every hot-path launch class: dispatches of the timed region grouped by (kernel, grid, LDS)."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [int(r["Start_Timestamp"]) for r in rows if _step_start(r["Kernel_Name"])]
    t0 = starts[warmup]
    steps = len(starts) - warmup
    agg = collections.defaultdict(list)
    for r in rows:
        if int(r["Start_Timestamp"]) >= t0 and ("lora_" in r["Kernel_Name"] or "attn_" in r["Kernel_Name"] or "geglu" in r["Kernel_Name"]):
            name = short(r["Kernel_Name"])[:46]
            key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1) or 1), int(r["LDS_Block_Size"]))
            agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':46s} {'blocks':>7s} {'gy':>4s} {'lds':>7s} {'n/step':>7s} {'avg us':>8s} {'min':>7s} {'max':>7s} {'us/step':>8s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:46s} {k[1]:7d} {k[2]:4d} {k[3]:7d} {len(v) / steps:7.1f} {sum(v) / len(v) / 1e3:8.1f} {min(v) / 1e3:7.1f} {max(v) / 1e3:7.1f} {sum(v) / steps / 1e3:8.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "shapes":
        shapes(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "trace":
        trace(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2:])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2], int(sys.argv[3]))
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
