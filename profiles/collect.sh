#!/bin/bash
# Collects the round's rocprofv3 evidence for every BASELINE config (run through gpurun from the repo root):
#   bash profiles/collect.sh r02 [configs, e.g. 25]
# Per config: kernel-trace + stats, then the PMC passes (separate runs), then the summaries that bench.py reads
# (profiles/<tag>/cfgN_pmc_summary.json, cfgN_kernel_stats.csv).
TAG=${1:-r02}
ONLY=${2:-2345}            # which configs, e.g. "25"
mkdir -p /root/repo/profiles/$TAG
for spec in "2|pmx_sw16_kernel<8, 19, 6>" "3|pmx_nwsg16q_kernel" "4|pmx_nwsg16v_kernel" "5|pmx_sw16_kernel<64, 16, 6>"; do
    case "$ONLY" in *"${spec%%|*}"*) ;; *) continue ;; esac
    set -- "${spec%%|*}" "${spec#*|}"
    bash /root/repo/profiles/run_profile.sh $TAG $1 "$2" > /root/repo/gpurun_out/collect_${TAG}_cfg$1.log 2>&1
    OUT=/root/repo/gpurun_out/prof_${TAG}_cfg$1
    cp $OUT/cfg$1_pmc_summary.json $OUT/cfg$1_kernel_stats.csv /root/repo/profiles/$TAG/ 2>/dev/null
    if [ "$1" = "4" ]; then python3 /root/repo/profiles/summarize_pmc.py $OUT pmx_walkp_kernel /root/repo/profiles/$TAG/cfg4_walk_pmc_summary.json 6; fi
    if [ "$1" = "3" ]; then python3 /root/repo/profiles/summarize_pmc.py $OUT pmx_walkp_kernel /root/repo/profiles/$TAG/cfg3_walk_pmc_summary.json 6; fi
    echo "cfg$1 done"; tail -3 /root/repo/gpurun_out/collect_${TAG}_cfg$1.log
done
cp -r /root/repo/profiles/$TAG /root/repo/gpurun_out/profiles_$TAG
