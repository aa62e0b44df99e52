#!/bin/bash
# Collect the judged evidence for one tag on a GPU box:
#   tools/profile_round.sh r1_c            (run from the repo root via gpurun)
# Writes gpurun_out/<tag>_*: bench JSON + layer table, rocprofv3 kernel stats,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE) the conv traffic figure comes from.
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
FSS_BENCH_LAYERS=$OUT/${TAG}_layers.txt timeout -k 10 300 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/${TAG}_trace.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency > $OUT/${TAG}_pmc_write.log 2>&1 || exit 4
python3 tools/summarize_pmc.py $OUT $TAG
