#!/bin/bash
# BatchNorm-backward tuning on the cfg-4 tensors: bench.py's operator timing (roofline.per_kernel) with the EXP=1 library and its knobs.
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS
    S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so python bench.py --config ${CFG:-cfg4} --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$label', d['ms_per_step'], ' '.join('%s=%.0fus/%.2f' % (k['kernel'].split(' ')[0]+k['kernel'].split('(')[1][:10], k['avg_us'], k['frac']) for k in r['per_kernel'][:4]))" )
done
