#!/bin/bash
# rocprofv3 kernel stats of one bench configuration:  tools/run_prof_cfg.sh cfg3 "--steps 3 --warmup 1"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
C=$1; S=${2:---steps 3 --warmup 1}
mkdir -p gpurun_out/r3
rm -rf gpurun_out/r3/prof_$C
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof_$C -o run -- python bench.py --config $C $S --no-cpu-baseline --no-roofline > gpurun_out/r3/prof_${C}_bench.json 2> gpurun_out/r3/prof_$C.err
DB=$(find gpurun_out/r3/prof_$C -name "*.db" | head -1)
python tools/prof_summary.py $DB > gpurun_out/r3/prof_${C}_summary.txt
find gpurun_out/r3/prof_$C -type f ! -name "*.txt" -delete
head -40 gpurun_out/r3/prof_${C}_summary.txt | cut -c1-190
