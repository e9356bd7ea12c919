#!/bin/bash
# rocprofv3 kernel table of one bench.py run: prof.sh <tag> [bench args...]   (environment passes through)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/r5
rm -rf gpurun_out/r5/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_$tag -o run -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline "$@" > gpurun_out/r5/prof_${tag}_bench.json 2> gpurun_out/r5/prof_$tag.err
DB=$(find gpurun_out/r5/prof_$tag -name "*.db" | head -1)
python tools/prof_summary.py $DB > gpurun_out/r5/${tag}_kernel_stats.txt
rm -rf gpurun_out/r5/prof_$tag
echo "== $tag"; head -${HEAD:-16} gpurun_out/r5/${tag}_kernel_stats.txt | cut -c1-170
