#!/bin/bash
# Collects the round's rocprofv3 evidence (run through gpurun from the repo root):
#   bash profiles/collect.sh r03 [which, e.g. "2345tl"]        2..5 = BASELINE configs, t = score tables, l = one long pair
# Per config, in this order:
#   1. bench.py --config N                      -> profiles/<tag>/bench_cfgN.json           (the pipelines as shipped: overlapped)
#   2. the same with the pipelines serialised   -> profiles/<tag>/bench_cfgN_serial.json    (cfg 3 / 4 only)
#   3. rocprofv3 --kernel-trace --stats of (2)  -> profiles/<tag>/cfgN_kernel_stats.csv     durations x launches <= ms_per_step of (2)
#   4. PMC passes of (2), each in its own run   -> profiles/<tag>/cfgN_pmc_summary.json     (read by bench.py, with the source blob)
# Launches that overlap on several streams have per-launch durations that cannot be added up (round-2 review), hence the
# serialised form for the trace; the overlapped line stands beside it.
TAG=${1:-r03}
ONLY=${2:-2345tl}
R=/root/repo
mkdir -p $R/profiles/$TAG $R/gpurun_out
export PYTHONUNBUFFERED=1
for spec in "2|pmx_sw16_kernel<8, 19|" "3|pmx_nwsg16q_kernel|pmx_walkp_kernel" "4|pmx_nwsg16v_kernel|pmx_walkp_kernel" "5|pmx_sw16_kernel<64, 16|pmx_banded"; do
    CFG=${spec%%|*}; REST=${spec#*|}; NEEDLE=${REST%%|*}; NEEDLE2=${REST#*|}
    case "$ONLY" in *"$CFG"*) ;; *) continue ;; esac
    python3 $R/bench.py --config $CFG > $R/profiles/$TAG/bench_cfg$CFG.json 2> $R/gpurun_out/bench_${TAG}_cfg$CFG.err || { echo "bench cfg$CFG failed"; tail -5 $R/gpurun_out/bench_${TAG}_cfg$CFG.err; exit 1; }
    echo "cfg$CFG bench done"
    if [ "$CFG" = "3" ] || [ "$CFG" = "4" ]; then
        export PMX_CIGAR_NO_OVERLAP=1 PMX_STATS_NO_OVERLAP=1
        python3 $R/bench.py --config $CFG --no-cpu-baseline > $R/profiles/$TAG/bench_cfg${CFG}_serial.json 2>> $R/gpurun_out/bench_${TAG}_cfg$CFG.err || exit 1
    fi
    bash $R/profiles/run_profile.sh $TAG cfg$CFG "$NEEDLE" 6 -- python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/collect_${TAG}_cfg$CFG.log 2>&1
    OUT=$R/gpurun_out/prof_${TAG}_cfg$CFG
    cp $OUT/cfg${CFG}_pmc_summary.json $OUT/cfg${CFG}_kernel_stats.csv $R/profiles/$TAG/ 2>/dev/null
    if [ -n "$NEEDLE2" ]; then python3 $R/profiles/summarize_pmc.py $OUT "$NEEDLE2" $R/profiles/$TAG/cfg${CFG}_${NEEDLE2#pmx_}_pmc_summary.json 6; fi
    unset PMX_CIGAR_NO_OVERLAP PMX_STATS_NO_OVERLAP
    echo "cfg$CFG done"; tail -3 $R/gpurun_out/collect_${TAG}_cfg$CFG.log
done
case "$ONLY" in *t*)
    bash $R/profiles/run_profile.sh $TAG table pmx_table_kernel 1 -- python3 $R/profiles/bench_tables.py > $R/gpurun_out/collect_${TAG}_table.log 2>&1
    cp $R/gpurun_out/prof_${TAG}_table/table_pmc_summary.json $R/gpurun_out/prof_${TAG}_table/table_kernel_stats.csv $R/profiles/$TAG/ 2>/dev/null
    echo "table done";;
esac
case "$ONLY" in *l*)
    bash $R/profiles/run_profile.sh $TAG long_single pmx_long 1 -- python3 $R/profiles/bench_long_single.py > $R/gpurun_out/collect_${TAG}_long.log 2>&1
    cp $R/gpurun_out/prof_${TAG}_long_single/long_single_pmc_summary.json $R/gpurun_out/prof_${TAG}_long_single/long_single_kernel_stats.csv $R/profiles/$TAG/ 2>/dev/null
    echo "long done";;
esac
cp -r $R/profiles/$TAG $R/gpurun_out/profiles_$TAG
