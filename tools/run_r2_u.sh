cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -3
bash tools/run_r2_t.sh > /dev/null 2>&1
python -c "import sys,json; d=json.loads(open('gpurun_out/r2/bench_final.json').read().strip().splitlines()[-1]); print('final', d['ms_per_step'], d['value'])"
