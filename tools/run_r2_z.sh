cd $GRAFT_REPO_ROOT
for i in 1 2; do
for f in 0 1; do S3D_PLANES_LAST=$f python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('PLANES_LAST=$f', d['ms_per_step'], d['value'])"; done
done
