cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
t() { python bench.py "$@" --no-cpu-baseline --no-roofline > gpurun_out/r5/f.log 2>&1; rc=$?; echo "rc=$rc  bench.py $*  -> $(grep '^{' gpurun_out/r5/f.log | python -c "import sys,json; l=sys.stdin.read().strip().splitlines(); d=json.loads(l[-1]) if l else {}; print(d.get('ms_per_step'), d.get('loss_first_step'), d.get('loss_last_step'))")"; [ $rc -ne 0 ] && grep -i "error" gpurun_out/r5/f.log | tail -2; true; }
t --config cfg4 --batch 40 --steps 10 --warmup 2
t --config cfg4 --batch 33 --steps 10 --warmup 2
t --config cfg4 --batch 64 --steps 10 --warmup 2
t --config cfg4 --batch 100 --steps 10 --warmup 2
t --config cfg4 --batch 130 --steps 10 --warmup 2
t --config cfg4 --batch 200 --steps 5 --warmup 2
t --config cfg5 --batch 17 --steps 10 --warmup 2
t --config cfg5 --batch 33 --steps 10 --warmup 2
t --config cfg5 --batch 64 --steps 5 --warmup 2
t --batch 100 --steps 50 --warmup 5
t --batch 320 --steps 20 --warmup 3
t --batch 1 --steps 50 --warmup 5
