cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
t() { python bench.py "$@" --no-cpu-baseline --no-roofline > gpurun_out/r5/f.log 2>&1; rc=$?; echo "rc=$rc  bench.py $*  -> $(grep '^{' gpurun_out/r5/f.log | python -c "import sys,json; l=sys.stdin.read().strip().splitlines(); print(json.loads(l[-1])['ms_per_step'] if l else 'no line')")"; [ $rc -ne 0 ] && grep -i "error" gpurun_out/r5/f.log | tail -2; }
t --steps 20 --warmup 3 --no-graphs
t --steps 20 --warmup 3 --plain-bf16
t --steps 50 --warmup 5 --force-collectives --wire bf16
t --steps 50 --warmup 5 --force-collectives --single-update
t --steps 50 --warmup 5 --force-collectives --graph-collectives off
t --steps 50 --warmup 5 --force-collectives --standin-gbps 140 --bucket-blocks 3
t --steps 50 --warmup 5 --force-collectives --event-graph
t --config cfg4 --steps 10 --warmup 2 --no-pipeline
t --config cfg4 --steps 10 --warmup 2 --no-graphs
t --config cfg5 --steps 10 --warmup 2 --force-collectives --no-pipeline
t --config cfg3 --steps 3 --warmup 1 --batch 16
t --steps 50 --warmup 5 --batch 8
