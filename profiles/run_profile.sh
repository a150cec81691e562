#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   profiles/run_profile.sh <tag> <config> [kernel substring for the PMC summary]
# kernel-trace + stats in one run; PMC counters in their own runs (never combined with traces).
# The program itself follows `--` (no launcher in between); the time limit wraps rocprofv3 from outside.
set -u
TAG=${1:-r02}
CFG=${2:-2}
NEEDLE=${3:-pmx_sw16_kernel}
OUT=/root/repo/gpurun_out/prof_${TAG}_cfg$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
if [ "${PMX_PROFILE_TRACE_ONLY:-0}" = "1" ]; then exit 0; fi
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
echo "pmc_sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o pmc -- $BENCH > $OUT/pmc_lds.log 2>&1
echo "pmc_lds rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?"
python3 /root/repo/profiles/summarize_pmc.py $OUT "$NEEDLE" $OUT/cfg${CFG}_pmc_summary.json 6
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/cfg${CFG}_kernel_stats.csv 2>/dev/null
ls $OUT
