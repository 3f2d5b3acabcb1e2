#!/bin/bash
# GPU box: SQ / LDS / traffic counters of the kernels whose name contains $1 (default resblock), over forward passes of $2 inputs (default 256):
# separate rocprofv3 --pmc passes (never together with a trace domain other than --kernel-trace).  Output: gpurun_out/probe_<name>.txt
M=${1:-resblock}; N=${2:-256}
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $G/pk_*
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $G/pk_1 -o x -- python $R/tools/probe_front.py $N 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $G/pk_2 -o x -- python $R/tools/probe_front.py $N 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $G/pk_3 -o x -- python $R/tools/probe_front.py $N 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $G/pk_4 -o x -- python $R/tools/probe_front.py $N 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $G/pk_5 -o x -- python $R/tools/probe_front.py $N 2 > /dev/null 2>&1
python $R/tools/pmc_sq.py $(find $G/pk_1 $G/pk_2 $G/pk_3 $G/pk_4 $G/pk_5 -name "x_results.db") --match $M > $G/probe_$M.txt 2>&1
rm -rf $G/pk_*
cat $G/probe_$M.txt
