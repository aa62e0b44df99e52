#!/bin/bash
# Collect the judged evidence for one tag on a GPU box:
#   tools/profile_round.sh r3_a [mixed|float32x3|bfloat16]     (run from the repo root via gpurun)
# Writes gpurun_out/<tag>_*: bench JSON + layer table, rocprofv3 kernel stats of the shipped
# (two-stream) graph and of the single-stream graph, the two TCC passes (FETCH_SIZE,
# WRITE_SIZE) the conv traffic figure comes from, and SQ / TCP / TD counter passes whose
# per-layer summary (MFMA pipe busy share, stalls) is <tag>_conv_pmc.txt.
# Every counter pass is a run of its own (no trace options beside --pmc).
set -o pipefail
TAG=${1:-r3}
DTYPE=${2:-mixed}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
QUICK="--quick --dtype $DTYPE"
FSS_BENCH_LAYERS=$OUT/${TAG}_layers.txt timeout -k 10 400 python3 bench.py --dtype $DTYPE --no-configs --no-throughput-mode --no-float32x3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 bench.py --steps 5 --warmup 2 $QUICK > $OUT/${TAG}_trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace1 -o t -- python3 bench.py --steps 5 --warmup 2 --streams 1 $QUICK > $OUT/${TAG}_trace1.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 0 $QUICK > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 bench.py --steps 1 --warmup 0 $QUICK > $OUT/${TAG}_pmc_write.log 2>&1 || exit 4
python3 tools/summarize_pmc.py $OUT $TAG || exit 5
i=0
for grp in \
 "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum" ; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/${TAG}_pmc_g$i -o p -- python3 bench.py --steps 1 --warmup 0 $QUICK > $OUT/${TAG}_pmc_g$i.log 2>&1 || { echo "counter pass $i failed"; tail -5 $OUT/${TAG}_pmc_g$i.log; exit 6; }
done
python3 tools/summarize_conv_pmc.py $OUT $TAG
