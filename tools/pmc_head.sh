#!/bin/bash
# PMC passes over one bench.py step, summarised for the classifier-head conv launch
# (conv_igemm_kernel<..., true, true>):  tools/pmc_head.sh [outdir]
set -o pipefail
OUT=${1:-gpurun_out/pmc_head}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in \
 "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 bench.py --steps 1 --warmup 1 --quick > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/g$i.log; exit 1; }
  echo "pass $i done"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for path in sorted(glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'conv_igemm_kernel' in r['Kernel_Name'] and 'true, true>' in r['Kernel_Name']:
                per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        acc[k] = sum(v[-2:]) / len(v[-2:])
for k, v in acc.items():
    print('%-32s %16.0f' % (k, v))
PY
find $OUT -name "*.csv" -size +1M -delete
