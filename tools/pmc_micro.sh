#!/bin/bash
# PMC passes over tools/conv_micro.py:  [DTYPE=bf16|f16|f32x3] tools/pmc_micro.sh <layer> <cfg> [batch]
# One rocprofv3 run per counter group; prints per-dispatch averages for the conv kernel.
set -o pipefail
L=${1:-cff1_dil}; C=${2:-3}; B=${3:-16}
OUT=gpurun_out/pmc_${L}_${C}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in \
 "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_SMEM" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 tools/conv_micro.py $L --cfg $C --batch $B --iters 3 --dtype ${DTYPE:-bf16} > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/g$i.log; exit 1; }
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
    print('%-32s %16.0f' % (k, v))
PY
