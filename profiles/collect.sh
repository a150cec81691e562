#!/bin/bash
# Collects the round's rocprofv3 evidence (run through gpurun from the repo root):
#   bash profiles/collect.sh r03 [which, e.g. "2345tl"]        2..5 = BASELINE configs, t = score tables, l = one long pair
# Per config, in this order:
#   1. rocprofv3 --kernel-trace --stats of the SERIALISED pipeline (cfg 3 / 4: PMX_*_NO_OVERLAP) -> cfgN_kernel_stats.csv, and the
#      JSON line that very run printed -> bench_cfgN_traced_run.json: durations x launches <= its ms_per_step
#   2. PMC passes of the same command, each in its own run -> cfgN_pmc_summary.json (read by bench.py, with the source blob id)
#   3. bench.py --config N serialised -> bench_cfgN_serial.json (cfg 3 / 4), and as shipped (overlapped) -> bench_cfgN.json
# Launches that overlap on several streams have per-launch durations that cannot be added up (round-2 review), hence the
# serialised form for the trace; the overlapped line stands beside it.
TAG=${1:-r03}
ONLY=${2:-2345tl}
R=/root/repo
mkdir -p $R/profiles/$TAG $R/gpurun_out
export PYTHONUNBUFFERED=1
for spec in "2|pmx_sw16_kernel<8, 19, 6>|" "3|pmx_nwsg16q_kernel|pmx_walkp_kernel" "4|pmx_nwsg16v_kernel|pmx_walkp_kernel" "5|pmx_sw16_kernel<64, 16, 6>|pmx_bstrip_kernel<8, 13, 3"; do
    CFG=${spec%%|*}; REST=${spec#*|}; NEEDLE=${REST%%|*}; NEEDLE2=${REST#*|}
    case "$ONLY" in *"$CFG"*) ;; *) continue ;; esac
    # (a) the serialised pipeline under rocprofv3: kernel trace, then the PMC passes; the traced run's own bench line is kept
    if [ "$CFG" = "3" ] || [ "$CFG" = "4" ]; then export PMX_CIGAR_NO_OVERLAP=1 PMX_STATS_NO_OVERLAP=1; fi
    bash $R/profiles/run_profile.sh $TAG cfg$CFG "$NEEDLE" 6 -- python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/collect_${TAG}_cfg$CFG.log 2>&1
    OUT=$R/gpurun_out/prof_${TAG}_cfg$CFG
    cp $OUT/cfg${CFG}_pmc_summary.json $OUT/cfg${CFG}_kernel_stats.csv $R/profiles/$TAG/ 2>/dev/null
    grep -h '^{"metric"' $OUT/trace.log | tail -1 > $R/profiles/$TAG/bench_cfg${CFG}_traced_run.json
    if [ -n "$NEEDLE2" ]; then
        NAME2=$(echo "${NEEDLE2#pmx_}" | sed 's/_kernel.*//')       # (a needle may carry template arguments: "pmx_bstrip_kernel<8, 13, 3")
        python3 $R/profiles/summarize_pmc.py $OUT "$NEEDLE2" "$R/profiles/$TAG/cfg${CFG}_${NAME2}_pmc_summary.json" 6
    fi
    # (b) the bench lines (they quote the counters just collected): serialised, then as shipped
    if [ "$CFG" = "3" ] || [ "$CFG" = "4" ]; then
        python3 $R/bench.py --config $CFG --no-cpu-baseline > $R/profiles/$TAG/bench_cfg${CFG}_serial.json 2>> $R/gpurun_out/bench_${TAG}_cfg$CFG.err || exit 1
    fi
    unset PMX_CIGAR_NO_OVERLAP PMX_STATS_NO_OVERLAP
    python3 $R/bench.py --config $CFG > $R/profiles/$TAG/bench_cfg$CFG.json 2> $R/gpurun_out/bench_${TAG}_cfg$CFG.err || { echo "bench cfg$CFG failed"; tail -5 $R/gpurun_out/bench_${TAG}_cfg$CFG.err; exit 1; }
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
