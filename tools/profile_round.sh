#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats and the PMC passes behind profiles/.
# Usage: bash tools/profile_round.sh [c3|train|all]   (outputs under gpurun_out/; then, in the build container,
#        python tools/summarise_pmc.py r04   copies the summaries into profiles/ under that round's name)
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa traces).  One gpurun call holds at
# most 20 minutes: `c3` (the headline frame: bench line, kernel stats, four PMC passes) and `train` (C4 / C5 / nerf step:
# bench line, kernel stats, three PMC passes each) fit one call each.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
part=${1:-all}
if [ "$part" != train ]; then
timeout -k 10 400 python3 bench.py --steps 5 --warmup 1 2>&1 | tail -1 > gpurun_out/bench.log || exit 1
cut -c1-600 gpurun_out/bench.log
rm -rf gpurun_out/prof_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame64 --no-train > gpurun_out/prof_stats.log 2>&1 || exit 1
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
    d=$(echo $c | tr " " "_" | cut -c1-40)
    rm -rf gpurun_out/pmc_$d
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$d -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-frame64 --no-train > gpurun_out/pmc_$d.log 2>&1 || exit 1
    echo "pmc $d ok"
done
fi
[ "$part" = c3 ] && exit 0
# training workloads: kernel stats + HBM / MFMA-busy counters of one step each (C4 / C5 = pi_GAN steps, nerf 1024-ray step)
for wl in c4 c5 nerf_train; do
    # the line that goes into profiles/ is an unprofiled run (rocprofv3 inflates the short kernels of the nerf step)
    if [ $wl = nerf_train ]; then st="--steps 40 --warmup 8"; else st="--steps 3 --warmup 1"; fi
    timeout -k 10 300 python3 bench.py --workload $wl $st 2>&1 | tail -1 > gpurun_out/bench_$wl.log || exit 1
    rm -rf gpurun_out/prof_stats_$wl
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 > gpurun_out/prof_stats_$wl.log 2>&1 || exit 1
    for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        d=${wl}_$(echo $c | tr " " "_" | cut -c1-30)
        rm -rf gpurun_out/pmct_$d
        timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_$d -- python3 bench.py --workload $wl --steps 1 --warmup 1 > gpurun_out/pmct_$d.log 2>&1 || exit 1
        echo "pmc $d ok"
    done
done
