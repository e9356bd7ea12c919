cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
# 1. the default bench line (what the driver runs) + the same command under rocprofv3 --kernel-trace --stats
python bench.py > gpurun_out/r2/bench_final.json 2> gpurun_out/r2/bench_final.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_final -o run -- python bench.py --no-cpu-baseline > gpurun_out/r2/bench_final_under_rocprof.json 2> /dev/null
# 2. PMC passes (separate runs, kernel-trace only)
bash tools/pmc_step.sh $PWD/gpurun_out/r2/pmcstep
ls gpurun_out/r2/pmcstep | head
tail -c 600 gpurun_out/r2/bench_final.json
