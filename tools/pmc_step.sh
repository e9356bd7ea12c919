#!/bin/bash
# HBM-side traffic and MFMA occupancy of every kernel of the cfg-2 training step (run on the GPU box):  tools/pmc_step.sh outdir
# Separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section); eager launches so that every kernel is
# a dispatch of its own.  tools/pmc_step_summary.py turns the CSVs into profiles/r01_pmc_step_traffic.json.
# PMC_BENCH_ARGS: other configuration, e.g. PMC_BENCH_ARGS="--config cfg3 --steps 1 --warmup 1" -> profiles/r03_pmc_cfg3_traffic.json
OUT=$1; R=$PWD; A=${PMC_BENCH_ARGS:---steps 4 --warmup 2}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $R/bench.py --no-graphs $A --no-cpu-baseline --no-roofline > $OUT/$1.log 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES"
cd $R
