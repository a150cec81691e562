#!/bin/bash
# Collects the rocprofv3 evidence for one command on the GPU box (run through gpurun):
#   profiles/run_profile.sh <tag> <label> <kernel substring> <launch divisor> -- <program and arguments>
# e.g. profiles/run_profile.sh r03 cfg4 pmx_nwsg16v_kernel 6 -- python3 /root/repo/bench.py --config 4 --steps 5 --warmup 1 --no-cpu-baseline
# kernel-trace + stats in one run; every PMC group in its own run (never combined with a trace); the program itself follows
# `--` (no launcher in between); the time limit wraps rocprofv3 from outside.  Environment switches (e.g. the serialised
# pipelines PMX_CIGAR_NO_OVERLAP / PMX_STATS_NO_OVERLAP) are exported by the caller and inherited.
# Output: gpurun_out/prof_<tag>_<label>/{<label>_kernel_stats.csv, <label>_pmc_summary.json, ...}
set -u
TAG=$1; LABEL=$2; NEEDLE=$3; DIV=$4; shift 4
[ "$1" = "--" ] && shift
OUT=/root/repo/gpurun_out/prof_${TAG}_${LABEL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${LABEL}_kernel_stats.csv 2>/dev/null
if [ "${PMX_PROFILE_TRACE_ONLY:-0}" = "1" ]; then exit 0; fi
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- "$@" > $OUT/pmc_sq.log 2>&1
echo "pmc_sq rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o pmc -- "$@" > $OUT/pmc_lds.log 2>&1
echo "pmc_lds rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- "$@" > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- "$@" > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?"
python3 /root/repo/profiles/summarize_pmc.py $OUT "$NEEDLE" $OUT/${LABEL}_pmc_summary.json $DIV
ls $OUT
