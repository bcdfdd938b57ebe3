#!/bin/bash
# GPU box: price the sin range-reduction variants (mi_math.h MI_SIN_VARIANT) - speed of every field kind in points mode
# (tools/perf_quick.py) and the parity records of the sin-family tests - one diagnostic library per variant, built
# beforehand in the build container with `bash tools/diag_build.sh sin1 sin2 sin3`.
#   bash tools/sin_variants_round.sh 0 1 2 3      (0 = the product library)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/sin
gpurun_tools/sin_variants > gpurun_out/sin/probe_accuracy.log 2>&1 || true
for k in "$@"; do
  if [ "$k" = 0 ]; then unset MI_DIAG_LIB; else export MI_DIAG_LIB=gpurun_tools/libmirender_sin$k.so; fi
  echo "== variant $k ==" | tee -a gpurun_out/sin/perf.log
  python tools/perf_quick.py 2>&1 | grep -E "fwd|train|using" | tee -a gpurun_out/sin/perf.log
  MI_PARITY_JSON=gpurun_out/sin/parity_sin$k.json timeout -k 10 900 python -m pytest -q -x -m gpu \
    tests/test_gpu_stages.py tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_c5.py \
    -k "siren or film or pigan or c5 or sloppier or field_golden or ragged" > gpurun_out/sin/tests_sin$k.log 2>&1 \
    && echo "variant $k tests: PASS" | tee -a gpurun_out/sin/perf.log \
    || { echo "variant $k tests: FAIL" | tee -a gpurun_out/sin/perf.log; tail -5 gpurun_out/sin/tests_sin$k.log; }
done
python - <<'PY'
import glob, json
for path in sorted(glob.glob("gpurun_out/sin/parity_sin*.json")):
    recs = json.load(open(path))["records"]
    worst = {}
    for r in recs:
        kind = "film" if "film" in r["case"] or "pigan" in r["case"].lower() or "C5" in r["case"] else "siren" if "siren" in r["case"] else None
        if kind is None or r.get("active") not in ("hard",) or not r.get("tol") or "err_vs_oracle32" not in r or r["qty"] not in ("rgb", "acc", "depth"):
            continue
        key = (kind, r["qty"])
        worst[key] = max(worst.get(key, 0.0), r["err_vs_oracle32"] / r["tol"])
    g = [r["rel_l2_err"] for r in recs if "rel_l2_err" in r]
    ge = [r["max_elem_err_over_rms"] for r in recs if "max_elem_err_over_rms" in r and "elem_tol" in r]
    print(path, {f"{k[0]}.{k[1]}": round(v, 4) for k, v in sorted(worst.items())}, "worst grad rel_l2", max(g, default=0), "worst grad elem", max(ge, default=0),
          "failed", sum(1 for r in recs if r.get("passed") is False))
PY
