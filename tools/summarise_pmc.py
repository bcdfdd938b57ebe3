"""Copy the rocprofv3 outputs of tools/profile_round.sh from gpurun_out/ into profiles/ and derive the
per-launch figures bench.py's roofline.traffic reads (profiles/rNN_pmc_nerf_fwd.json).

FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE under-reports by 2x (MI355X_MICROARCH.md, HBM / rocprofv3
section).  Both the raw and the corrected figures are written out."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "profiles")
KERNEL = "nerf_fwd_kernel<false, false>"


def newest(pattern):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


stats = newest("prof_stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(out_dir, f"{tag}_bench_c3_kernel_stats.csv"))
bench = os.path.join(ROOT, "gpurun_out", "bench.log")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(out_dir, f"{tag}_bench_n1.log"))

counters, launches = {}, None
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    shutil.copy(f, os.path.join(out_dir, f"{tag}_{os.path.basename(d)}_counter_collection.csv"))
    per = {}
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        key = int(r["Dispatch_Id"])
        per.setdefault(key, {"grid": int(r["Grid_Size"]), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        per[key][r["Counter_Name"]] = float(r["Counter_Value"])
    order = sorted(per)
    if launches is None:
        launches = [{"points": per[k]["grid"] // 256 * 128, "ms": per[k]["ns"] / 1e6} for k in order]
    for k in order:
        for name, v in per[k].items():
            if name not in ("grid", "ns"):
                counters.setdefault(name, []).append(v)

if not counters:
    sys.exit("no PMC passes found under gpurun_out/")
i = max(range(len(launches)), key=lambda j: launches[j]["points"])      # the fine pass: most points
pts, ms = launches[i]["points"], launches[i]["ms"]
g = lambda n: counters[n][i]
fetch, write = g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
derived = {
    "points": pts, "ms": ms,
    "WRITE_SIZE_bytes": write, "FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_bytes_x2_gfx950_correction": 2 * fetch,
    "hbm_bytes_per_point_upper": (2 * fetch + write) / pts,
    "hbm_GBps_upper": (2 * fetch + write) / (ms * 1e-3) / 1e9,
    "effective_clock_GHz": g("GRBM_GUI_ACTIVE") / 8 / (ms * 1e-3) / 1e9,
    # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs: SIMD-cycles = GRBM x 128
    "mfma_busy_frac": g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") * 128),
    "mfma_flops_issued": g("SQ_INSTS_VALU_MFMA_MOPS_F32") * 512,
    "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"),
    "wave_cycles_split": {n: g(n) / g("SQ_WAVE_CYCLES") for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")},
}
json.dump({
    "command": "bash tools/profile_round.sh (rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 "
               "bench.py --steps 1 --warmup 0 --no-cpu-baseline; one pass per counter group)",
    "kernel": "mi::nerf_fwd_kernel<false,false>",
    "launch_points": [l["points"] for l in launches], "launch_ms": [l["ms"] for l in launches],
    "counters": counters, "derived_fine_launch": derived}, open(os.path.join(out_dir, f"{tag}_pmc_nerf_fwd.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))


# ---- training workloads: per-kernel HBM traffic and MFMA-busy fraction over the traced run (profile_round.sh) ----
def train_summary(wl):
    per_kernel = {}
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmct_{wl}_*"))):
        if not os.path.isdir(d):
            continue
        f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not name.startswith("mi::"):
                continue
            k = per_kernel.setdefault(name, {"launches": 0, "ms": 0.0})
            if r["Counter_Name"] == "FETCH_SIZE":                 # one row per dispatch in that pass: count and time it
                k["launches"] += 1
                k["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for name, k in per_kernel.items():
        if not k.get("ms"):
            continue
        hbm = (2 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0)) * 1024
        row = {"launches": k["launches"], "ms": k["ms"], "hbm_bytes_2xFETCH_plus_WRITE": hbm,
               "hbm_GBps": hbm / (k["ms"] * 1e-3) / 1e9}
        if k.get("GRBM_GUI_ACTIVE"):
            row["mfma_busy_frac"] = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (k["GRBM_GUI_ACTIVE"] * 128)
        out[name] = row
    stats = newest(f"prof_stats_{wl}/**/*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(out_dir, f"{tag}_bench_{wl}_kernel_stats.csv"))
    log = os.path.join(ROOT, "gpurun_out", f"bench_{wl}.log")              # the unprofiled run of tools/profile_round.sh
    if not os.path.exists(log):
        log = os.path.join(ROOT, "gpurun_out", f"prof_stats_{wl}.log")
    if os.path.exists(log):
        line = [l for l in open(log) if l.startswith("{")]
        if line:
            open(os.path.join(out_dir, f"{tag}_bench_{wl}.log"), "w").write(line[-1])
    if out:
        settle = 2 if wl in ("c4", "c5") else 0       # bench.py's allocator-settle steps run traced too
        steps_traced = settle + 1 + 1
        for row in out.values():
            row["hbm_bytes_per_step"] = row["hbm_bytes_2xFETCH_plus_WRITE"] / steps_traced
        json.dump({"command": f"rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --workload {wl} --steps 1 --warmup 1 "
                              f"(one pass per counter group; sums over every step that runs traced: {settle} allocator-settle "
                              "+ 1 warm-up + 1 timed)",
                   "steps_traced": steps_traced,
                   "hbm_bytes_per_step": sum(r["hbm_bytes_per_step"] for r in out.values()),
                   "note": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH doubled for gfx950 (MI355X_MICROARCH.md)", "kernels": out},
                  open(os.path.join(out_dir, f"{tag}_pmc_{wl}.json"), "w"), indent=1)
        top = sorted(out.items(), key=lambda kv: -kv[1]["ms"])[:6]
        for name, row in top:
            print(f"{wl:10s} {name[:60]:60s} {row['ms']:9.2f} ms {row['hbm_GBps']:8.1f} GB/s  busy {row.get('mfma_busy_frac', float('nan')):.3f}")


for wl in ("c4", "c5", "nerf_train"):
    train_summary(wl)
