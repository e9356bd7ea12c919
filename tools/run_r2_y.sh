cd $GRAFT_REPO_ROOT
for i in 1 2; do
for l in libs3d_hip_prev.so libs3d_hip.so; do S3D_LIB_PATH=$PWD/simple3d-former_amd/$l python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['ms_per_step'], d['value'])"; done
done
