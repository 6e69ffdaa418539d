"""Boil the rocprofv3 output of tools/profile_round.sh down to the small files kept under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s" % tag
dst = "gpurun_out/profiles_%s" % tag
os.makedirs(dst, exist_ok=True)


def norm(name):
    """'void k_stack_bwd<8>(TrainParams, TrainBwd, StackQ)' -> 'k_stack_bwd<8>': the name bench.py's roofline.kernels carry"""
    name = name.split("(")[0].strip()
    return (name[5:] if name.startswith("void ") else name)[:80]


def find(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


for what in ("train", "train2", "decode", "default", "default_dec"):
    f = find("%s_stats/**/*kernel_stats.csv" % what)
    if f:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, what)), "w") as o:
            w = csv.writer(o, quoting=csv.QUOTE_ALL)
            for r in rows[:40]:
                r[0] = r[0][:160]
                w.writerow(r)

traffic = {"note": "HBM bytes = 2*FETCH_SIZE + WRITE_SIZE (KB*1024; gfx950 correction of MI355X_MICROARCH.md), separate --pmc passes, summed per kernel name"}
per = {}
for what in ("train", "decode"):
    agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = find("%s_%s/**/*counter_collection.csv" % (what, c))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
            name = norm(name)
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE":
                calls[name] += 1
    per[what] = {k: {"calls": calls[k], "FETCH_SIZE_KB": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0.0),
                     "hbm_bytes": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024} for k, v in agg.items()}
# matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES summed over a dispatch / (its duration x clock x SIMDs)
f = find("train_MFMA/**/*counter_collection.csv")
fs = find("train_stats/**/*kernel_stats.csv")
if f and fs:
    avg_ns = {}
    for r in csv.DictReader(open(fs)):
        avg_ns[norm(r["Name"])] = float(r["AverageNs"])
    busy = defaultdict(float); n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        name = norm(r.get("Kernel_Name") or "")
        busy[name] += float(r["Counter_Value"]); n[name] += 1
    for name in busy:
        if name in avg_ns and name in per.get("train", {}):
            per_call = busy[name] / n[name]
            per["train"][name]["mfma_busy_cycles_per_call"] = per_call
            per["train"][name]["avg_us"] = avg_ns[name] / 1e3
            per["train"][name]["mfma_util"] = per_call / (avg_ns[name] * 2.4 * 1024)
            per["train"][name]["mfma_busy"] = round(per["train"][name]["mfma_util"], 3)        # (what bench.py's roofline.kernels[].mfma_busy reads)
# where the waves' cycles go (SQ wait / active counters), per kernel
f = find("train_WAIT/**/*counter_collection.csv")
if f:
    agg = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[norm(r.get("Kernel_Name") or "")][r["Counter_Name"]] += float(r["Counter_Value"])
    with open(os.path.join(dst, "%s_train_wait_pmc.txt" % tag), "w") as o:
        o.write("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU\n")
        o.write("  -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu   (QPN_TRAIN_SERIAL=1); fractions of SQ_WAVE_CYCLES summed over a kernel's dispatches\n")
        o.write("%-44s %10s %8s %8s %8s %8s %8s %8s %8s\n" % ("kernel", "wave_cyc", "wait", "waitinst", "active", "wlds", "alds", "avmem", "avalu"))
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:16]:
            w = v["SQ_WAVE_CYCLES"] or 1
            o.write("%-44s %10.3g %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n" % (k[:44], w, v["SQ_WAIT_ANY"] / w, v["SQ_WAIT_INST_ANY"] / w, v["SQ_ACTIVE_INST_ANY"] / w,
                                                                                   v["SQ_WAIT_INST_LDS"] / w, v["SQ_ACTIVE_INST_LDS"] / w, v["SQ_ACTIVE_INST_VMEM"] / w, v["SQ_ACTIVE_INST_VALU"] / w))
json.dump(per, open(os.path.join(dst, "%s_pmc_by_kernel.json" % tag), "w"), indent=1)
# the figures bench.py reports
dec = [v for k, v in per.get("decode", {}).items() if "k_decode" in k]
if dec:
    frames = 2005 if tag != "r01" else 600
    n = 20 * (frames * 110 - 1)
    names = [k for k in per.get("decode", {}) if "k_decode" in k]
    traffic["decode"] = {"kernel": ", ".join(names), "workload": "batch 20 x %d frames (20 x %d samples)" % (frames, frames * 110 - 1),
                         "hbm_bytes_per_sample": sum(v["hbm_bytes"] for v in dec) / n}
wg = [(k, v) for k, v in per.get("train", {}).items() if "k_wgrad" in k]
if wg:
    steps = 3.0    # 1 warmup + 2 timed steps in the PMC runs (the profiled extra steps of bench.py are included: see calls)
    calls = max(v["calls"] for _, v in wg)
    nsteps = max(per["train"].get("k_train_prep", {}).get("calls", 0), 1)          # steps of the profiled command (timed + warm-up + bench.py's per-group steps)
    GROUP = (("k_train_prep", "prep+pack"), ("k_refresh", "prep+pack"), ("k_aux_tail", "grad_tail"), ("k_stack_fwd", "k_layer_fwd"), ("k_layer_fwd", "k_layer_fwd"), ("k_post_fwd", "k_post_fwd"),
             ("k_ce", "k_ce"), ("k_post_bwd", "k_post_bwd"), ("k_wgrad", "k_wgrad"), ("k_stack_bwd", "k_layer_bwd"), ("k_layer_bwd", "k_layer_bwd"),
             ("k_reduce_grad", "grad_tail"), ("k_up_bwd", "grad_tail"), ("k_causal_bwd", "grad_tail"), ("k_zero_dx", "k_post_bwd"), ("k_adam", "k_adam"))
    by_group = defaultdict(float)
    for k, v in per["train"].items():
        for key, grp in GROUP:
            if key in k:
                by_group[grp] += v["hbm_bytes"] / nsteps
                break
    traffic["train"] = {"kernel": "k_wgrad3 (5 launches per step)", "workload": "paper-size step, chunk 20900 samples (RF 946 + 19954), QPN_TRAIN_SERIAL=1",
                        "steps_profiled": nsteps,
                        "hbm_bytes_per_step": by_group.get("k_wgrad", 0.0),
                        "hbm_bytes_by_group": {k: round(v) for k, v in by_group.items()},
                        "hbm_bytes_by_kernel": {k: round(v["hbm_bytes"] / max(v["calls"], 1)) for k, v in per["train"].items() if k.startswith("k_")},      # per LAUNCH
                        "hbm_bytes_per_step_all_kernels": round(sum(v["hbm_bytes"] for k, v in per["train"].items() if k.startswith("k_")) / nsteps)}
traffic["commit"] = os.environ.get("QPN_COMMIT", "?")
json.dump(traffic, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
print(json.dumps(traffic, indent=1))
