#!/bin/bash
# SQ counters of the stem kernel:  tools/pmc_stem.sh [layer]
set -o pipefail
L=${1:-branch_conv}
OUT=gpurun_out/pmc_stem_${L}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SMEM" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 tools/stem_micro.py $L --iters 3 --batch ${BATCH:-16} > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/g$i.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for path in sorted(glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'stem_conv_kernel' in r['Kernel_Name'] or 'stem2_conv_kernel' in r['Kernel_Name']:
                per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        acc[k] = sum(v[-3:]) / len(v[-3:])
for k, v in acc.items():
    print('%-32s %16.0f' % (k, v))
PY
