#!/bin/bash
# Collects the rocprofv3 evidence for one precision on the GPU box (run through gpurun):
#   bash tools/profile_round.sh f16x3|f32
# kernel stats and each PMC group are separate runs (the counters do not fit one pass, and --pmc
# must not be combined with other trace domains); summarise afterwards with tools/pmc_summary.py.
set -e
PREC=${1:-f16x3}
OUT=gpurun_out/prof_$PREC
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --steps 5 --warmup 1 --no-alt --no-cpu-baseline --no-full-loop --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_stats -- python3 $ARGS > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p_fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p_write -- python3 $ARGS > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p_sq -- python3 $ARGS > $OUT/sq.log 2>&1
echo "sq done"
tail -1 $OUT/stats.log
