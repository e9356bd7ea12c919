#!/bin/bash
# PMC passes over one eager cfg-3 step (run on the GPU box):  tools/pmc_cfg3.sh outdir
OUT=$1; R=$PWD
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { timeout 600 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $R/bench.py --config cfg3 --no-graphs --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM"
run sq3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run fetch "FETCH_SIZE"
cd $R
