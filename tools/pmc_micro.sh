#!/bin/bash
# SQ counter passes over a micro-benchmark script (one rocprofv3 --pmc run per group): bash tools/pmc_micro.sh TAG script.py [args]
set -o pipefail
TAG=${1:-pm}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python $R/"$@" > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $O/g$i.log; exit 1; }
  find $O/g$i -name "*kernel_trace.csv" -delete
done
python $R/tools/pmc_micro_summary.py $O | tee $O/summary.txt
