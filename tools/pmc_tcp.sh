#!/bin/bash
# TCP / TA counters of the conv kernel on one layer:  tools/pmc_tcp.sh <layer> <cfg> [batch]
set -o pipefail
L=${1:-cff1_dil}; C=${2:-16}; B=${3:-16}
OUT=gpurun_out/tcp_${L}_${C}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in \
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 tools/conv_micro.py $L --cfg $C --batch $B --iters 3 > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/g$i.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for path in sorted(glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'conv_igemm_kernel' in r['Kernel_Name']:
                per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        acc[k] = sum(v[-3:]) / len(v[-3:])
for k, v in acc.items():
    print('%-40s %16.0f' % (k, v))
PY
