#!/bin/bash
# Diagnostic: PMC groups for the kernels of one command, printed per kernel (mean per dispatch).
# Usage (GPU box): bash tools/probes/pmc_kernel.sh "<counters group 1>" "<group 2>" ... -- python3 tools/perf_train_nerf.py
export TMPDIR=/tmp
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
i=0
for g in "${groups[@]}"; do
  d=gpurun_out/pmck_$i; rm -rf $d
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- "$@" > $d.log 2>&1 || { echo "pass $i failed"; tail -3 $d.log; exit 1; }
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("mi::") or not any(s in k for s in ("fwd_kernel", "bwd_kernel", "dw_gemm_kernel<4, 2, 2>")):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in n[k]:
        n[k].add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k in acc:
    print(f"{k[:44]:44s} n={len(n[k]):3d} {dur[k]/len(n[k]):8.3f} ms  " + "  ".join(f"{c}={v/len(n[k]):.4g}" for c, v in sorted(acc[k].items())))
PY
  i=$((i+1))
done
