#!/usr/bin/env python3
"""Every figure DESIGN.md / README.md quote for the headline kernel and the bench line, asserted against the tracked file
under profiles/ it is said to come from (the docs quoted 94.0 % / 994.6 ms / 22.4 B/point / 6 208 MFMAs where the
tracked profiles said 0.936 / 1 001.2 ms / 21.2 B/point / 9 280).

    python tools/check_figures.py            # prints one line per figure, exits non-zero on a mismatch or a missing sentence

A check = (document, regular expression with one group per number, the values the tracked files give, tolerance).  The
sentence must be FOUND - a figure cannot leave the check by being reworded - and every captured number must equal the
source's within the rounding the text shows (half a unit of its last printed digit, plus `slack` relative).  CPU only;
tests/test_figures.py runs it in the CPU suite."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r04"
PEAK = 157.3


def _json_line(path):
    with open(os.path.join(ROOT, path)) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def _num(text: str) -> float:
    return float(text.replace(" ", "").replace(" ", "").replace(",", ""))


def _digits(text: str) -> float:
    """Half a unit of the last digit the text prints ('0.930' -> 0.0005, '482 252' -> 0.5, '3.5635e10' -> 0.00005e10)."""
    t = text.replace(" ", "")
    mant, _, exp = t.lower().partition("e")
    scale = 10.0 ** int(exp) if exp else 1.0
    frac = len(mant.split(".")[1]) if "." in mant else 0
    return 0.5 * 10.0 ** (-frac) * scale


def sources():
    bench = _json_line(f"profiles/{ROUND}_bench_n1.log")
    pmc = json.load(open(os.path.join(ROOT, f"profiles/{ROUND}_pmc_nerf_fwd.json")))
    d = pmc["derived_fine_launch"]
    i = max(range(len(pmc["launch_points"])), key=lambda j: pmc["launch_points"][j])
    waves = pmc["launch_points"][i] / 32
    stats = {}
    with open(os.path.join(ROOT, f"profiles/{ROUND}_bench_c3_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            stats[r["Name"]] = r
    fwd = next(v for k, v in stats.items() if "nerf_fwd_kernel<false, false>" in k)
    flops_fine = pmc["launch_points"][i] * 1182976
    kinds = json.load(open(os.path.join(ROOT, f"profiles/{ROUND}_pmc_kinds.json")))["kernels"]
    kb = lambda name: kinds[name]["mfma_busy_frac"]                                   # noqa: E731
    pct = {}
    with open(os.path.join(ROOT, f"profiles/{ROUND}_perf_kinds.log")) as f:
        for ln in f:
            m = re.match(r"(\w+)\s+fwd .*\(([\d.]+)% of fp32 MFMA peak\)", ln)
            if m:
                pct[m.group(1)] = float(m.group(2))
    fam = bench["psnr_vs_ref"]["families"]
    tr = bench["train"]
    return dict(
        bench=bench, pmc=d,
        fine_ms=d["ms"], fine_tflops=flops_fine / (d["ms"] * 1e-3) / 1e12, busy=d["mfma_busy_frac"], clock=d["effective_clock_GHz"],
        write_gb=d["WRITE_SIZE_bytes"] / 1e9, write_bpp=d["WRITE_SIZE_bytes"] / d["points"], fetch_gb=d["FETCH_SIZE_bytes_raw"] / 1e9,
        bpp=d["hbm_bytes_per_point_upper"], gbps=d["hbm_GBps_upper"], lds=d["lds_bank_conflict_cycles"],
        insts_mfma=pmc["counters"]["SQ_INSTS_MFMA"][i], mfma_per_wave=pmc["counters"]["SQ_INSTS_MFMA"][i] / waves,
        flops_issued=d["mfma_flops_issued"], flops_alg=flops_fine,
        stats_calls=int(fwd["Calls"]), stats_avg_ms=float(fwd["AverageNs"]) / 1e6, fam=fam, tr=tr, kb=kb, pct=pct)


def checks(s):
    b, fam, tr = s["bench"], s["fam"], s["tr"]
    r = b["roofline"]
    design = [
        # section 4.1: the PMC passes of the headline kernel
        (r"fine launch ([\d.]+) ms = ([\d.]+) TFLOP/s = ([\d.]+) % of the",
         [s["fine_ms"], s["fine_tflops"], 100 * s["fine_tflops"] / PEAK]),
        (r"`SQ_VALU_MFMA_BUSY_CYCLES` / SIMD-cycles = ([\d.]+); effective clock ([\d.]+) GHz; HBM traffic ([\d.]+) GB written",
         [s["busy"], s["clock"], s["write_gb"]]),
        (r"\(= ([\d.]+) B/point\) \+ ([\d.]+) GB x 2 read per launch = ([\d.]+) B/point against 20 algorithmic, ([\d.]+) GB/s",
         [s["write_bpp"], s["fetch_gb"], s["bpp"], s["gbps"]]),
        (r"`SQ_LDS_BANK_CONFLICT` (\d+); `SQ_INSTS_MFMA` ([\d.e]+) = ([\d ]+) MFMAs per wave-tile",
         [s["lds"], s["insts_mfma"], s["mfma_per_wave"]]),
        (r"MFMA FLOPs issued ([\d.e]+) against ([\d.e]+) algorithmic", [s["flops_issued"], s["flops_alg"]]),
        (r"`nerf_fwd_kernel<false,false>` (\d+) launches, average ([\d.]+) ms \(coarse \+ fine launch of a step = ([\d .]+) ms\)",
         [s["stats_calls"], s["stats_avg_ms"], 2 * s["stats_avg_ms"]]),
        (r"besides its ([\d ]+) MFMAs \(", [s["mfma_per_wave"]]),
        (r"`profiles/r04_perf_kinds.log`\): NeRF ([\d.]+) %, TinyNeRF ([\d.]+) %, SirenNeRF ([\d.]+) %, FilmSirenNeRF ([\d.]+) % of",
         [s["pct"]["nerf"], s["pct"]["tiny"], s["pct"]["siren"], s["pct"]["film"]]),
        (r"`profiles/r04_pmc_kinds.json`\): ([\d.]+) / ([\d.]+) / ([\d.]+) / ([\d.]+), and - a NeRF training step at 8 192 rays - saving forward ([\d.]+), chain ([\d.]+),\s+256x256 dW GEMMs ([\d.]+), the narrow tiles ([\d.]+) \(128x256\) / ([\d.]+) \(256x64\) / ([\d.]+) \(128x32\)",
         [s["kb"]("mi::nerf_fwd_kernel<false, false>"), s["kb"]("mi::nerf_fwd_kernel<true, false>"), s["kb"]("mi::siren_fwd_kernel<false>"),
          s["kb"]("mi::film_fwd_kernel<true, false>"), s["kb"]("mi::nerf_fwd_kernel<false, true>"), s["kb"]("mi::nerf_bwd_kernel<false>"),
          s["kb"]("mi::dw_gemm_kernel<4, 2, 2>"), s["kb"]("mi::dw_gemm_kernel<4, 1, 2>"), s["kb"]("mi::dw_gemm_kernel<2, 2, 1>"),
          s["kb"]("mi::dw_gemm_kernel<1, 1, 1>")]),
        # section 6: the bench line
        (r"one MI355X\): ([\d ]+) rays/s, ([\d .]+) ms per frame, `roofline.frac` ([\d.]+)\s+\(([\d.]+) TFLOP/s; average MLP launch ([\d.]+) ms",
         [b["value"], b["ms_per_step"], r["frac"], r["achieved"], r["avg_launch_ms"]]),
        (r"`frame64` \(800x800 @ 64 samples\) ([\d.]+) ms = ([\d.]+) M rays/s", [b["frame64"]["ms_per_frame"], b["frame64"]["rays_per_s"] / 1e6]),
        (r"same frame\) (\d+) rays/s, GPU/CPU (\d+)x", [b["cpu_baseline"]["value"], b["gpu_over_cpu"]]),
        (r"FETCH_SIZE doubled as the guide prescribes for\s+gfx950\): ([\d.]+) B/point against 20 algorithmic", [s["bpp"]]),
        (r"TinyNeRF ([\d.]+) vs ([\d.]+) dB \(oracle loop run live\), loss ([\d.e-]+)",
         [fam["tiny_nerf"]["hip_db"], fam["tiny_nerf"]["cpu_reference_loop_db"], fam["tiny_nerf"]["max_rel_loss_diff"]]),
        (r"round 4 - ([\d.]+) vs ([\d.]+) dB, loss ([\d.e-]+); SirenNeRF ([\d.]+) vs ([\d.]+) dB, loss ([\d.e-]+); FilmSirenNeRF \(fixed FiLM row\)\s+([\d.]+) vs ([\d.]+) dB, loss ([\d.e-]+)",
         [fam["nerf"]["hip_db"], fam["nerf"]["cpu_reference_loop_db"], fam["nerf"]["max_rel_loss_diff"],
          fam["siren_nerf"]["hip_db"], fam["siren_nerf"]["cpu_reference_loop_db"], fam["siren_nerf"]["max_rel_loss_diff"],
          fam["film_siren_nerf"]["hip_db"], fam["film_siren_nerf"]["cpu_reference_loop_db"], fam["film_siren_nerf"]["max_rel_loss_diff"]]),
        (r"nerf 1 024-ray step ([\d.]+) ms, `frac` ([\d.]+); pi_GAN C5 step \(4 images 256x256, D-step forward \+\s+G-step\) ([\d.]+) ms, `frac` ([\d.]+) \(reference-equivalent ([\d.]+)\); C4 step \(batch 32, 128x128\) ([\d.]+) ms, `frac` ([\d.]+)\s+\(reference-equivalent ([\d.]+)\)",
         [tr["nerf_train"]["ms_per_step"], tr["nerf_train"]["frac"], tr["c5"]["ms_per_step"], tr["c5"]["frac"], tr["c5"]["frac_reference_equivalent"],
          tr["c4"]["ms_per_step"], tr["c4"]["frac"], tr["c4"]["frac_reference_equivalent"]]),
    ]
    readme = [
        (r"profiles/r04_\*\): ([\d ]+) rays/s on the 64\+128 frame \(([\d.]+) s per 800×800 frame\), ([\d.]+) ms per\s+800×800 frame at 64 samples",
         [b["value"], b["ms_per_step"] / 1e3, b["frame64"]["ms_per_frame"]]),
        (r"fused NeRF MLP at ([\d.]+) TFLOP/s = ([\d.]+) % of the fp32 MFMA peak \(MFMA-busy counter ([\d.]+)\)",
         [r["achieved"], 100 * r["frac"], s["busy"]]),
        (r"(\d+)× the 16-core CPU oracle", [b["gpu_over_cpu"]]),
        (r"nerf\s+step \(1024 rays, fused Adam\) ([\d.]+) ms \(([\d.]+) of peak\)", [tr["nerf_train"]["ms_per_step"], tr["nerf_train"]["frac"]]),
        (r"generator\s+step ([\d.]+) s \(([\d.]+) of peak on the 108", [tr["c4"]["ms_per_step"] / 1e3, tr["c4"]["frac"]]),
        (r"256² training step ([\d.]+) s per GPU \(([\d.]+)\)", [tr["c5"]["ms_per_step"] / 1e3, tr["c5"]["frac"]]),
    ]
    return [("DESIGN.md", design), ("README.md", readme)]


def _format_like(text: str, value: float) -> str:
    """`value` printed the way `text` prints its number: same decimals, thin-space thousands, mantissa digits of an e-notation."""
    t = text.strip()
    if "e" in t.lower():
        mant, _, exp = t.lower().partition("e")
        frac = len(mant.split(".")[1]) if "." in mant else 0
        e = int(exp)
        out = f"{value / 10.0 ** e:.{frac}f}e{exp}"
        return out
    frac = len(t.split(".")[1]) if "." in t else 0
    out = f"{value:.{frac}f}"
    if " " in t or "\u2009" in t:                       # grouped thousands ("482 252")
        whole, dot, rest = out.partition(".")
        groups = []
        while len(whole) > 3:
            groups.insert(0, whole[-3:])
            whole = whole[:-3]
        out = " ".join([whole] + groups) + dot + rest
    return out


def fix():
    """Rewrite every checked number in DESIGN.md / README.md from the tracked files, keeping each number's printed format
    (after a new profiling round: `python tools/summarise_pmc.py rNN && python tools/check_figures.py --fix`)."""
    s = sources()
    for doc, items in checks(s):
        path = os.path.join(ROOT, doc)
        text = open(path).read()
        for pattern, want in items:
            m = re.search(pattern, text)
            if not m:
                print(f"{doc}: sentence not found: /{pattern[:90]}.../")
                continue
            piece = m.group(0)
            new, last = "", 0
            for gi, w in enumerate(want, start=1):
                a, b = m.start(gi) - m.start(0), m.end(gi) - m.start(0)
                new += piece[last:a] + _format_like(m.group(gi), w)
                last = b
            new += piece[last:]
            text = text[:m.start(0)] + new + text[m.end(0):]
        open(path, "w").write(text)


def main(verbose=True, slack=2e-4):
    s = sources()
    failures = []
    for doc, items in checks(s):
        text = open(os.path.join(ROOT, doc)).read()
        for pattern, want in items:
            m = re.search(pattern, text)
            if not m:
                failures.append(f"{doc}: sentence not found: /{pattern[:90]}.../")
                continue
            for got_text, w in zip(m.groups(), want):
                got = _num(got_text)
                tol = _digits(got_text) * 1.02 + slack * abs(w)
                ok = abs(got - w) <= tol
                if verbose:
                    print(f"{'ok ' if ok else 'BAD'} {doc:10s} quoted {got_text.strip():>12s}  tracked {w:.6g}")
                if not ok:
                    failures.append(f"{doc}: quotes {got_text.strip()} where the tracked profiles give {w:.6g} (/{pattern[:60]}.../)")
    return failures


if __name__ == "__main__":
    if "--fix" in sys.argv:
        fix()
    bad = main(verbose="--fix" not in sys.argv)
    for f in bad:
        print("MISMATCH:", f)
    sys.exit(1 if bad else 0)
