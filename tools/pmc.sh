#!/bin/bash
# PMC counter collection for one GEMM of tools/gemm_bench.py (run on the GPU box):  tools/pmc.sh fwd_fc1 outdir
# Separate passes (SQ block has 8 slots, TCC 4); --kernel-trace only, as the guide prescribes.
ONLY_SEL=$1; OUT=$2; R=$PWD
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { ONLY=$ONLY_SEL rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $R/tools/gemm_bench.py > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM"
run sq3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run tcc2 "FETCH_SIZE"
run tcc3 "WRITE_SIZE"
cd $R
