cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2/prof_a
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_a -o run -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r2/prof_a_bench.json 2> gpurun_out/r2/prof_a.err
ls -R gpurun_out/r2/prof_a | head -20
python tools/prof_summary.py gpurun_out/r2/prof_a > gpurun_out/r2/prof_a_summary.txt 2>&1
head -40 gpurun_out/r2/prof_a_summary.txt | cut -c1-200
