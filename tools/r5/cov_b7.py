# which GEMM instantiations does a cfg-3 B=7 step launch? (prints the fat-tile families)
import sys; sys.path.insert(0, '.')
import torch, ctypes
import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from oracle import voxel_oracle as vo
from tests import _cov as C
kw = dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
sd = vo.init_state_dict(seed=9, pos_embedding='group_embed', **kw)
x, y = vo.synthetic_batch(7, 128, 55, seed=9)
eng = s3d.VoxelEngine(device='cuda', pos_embedding='group_embed', **kw); eng.load_state_dict(sd); eng.set_dropout(0.1, seed=5)
xd, yd = x.cuda(), y.cuda()
eng.train_step(xd, yd)
lib = L.lib(); lib.s3d_cov_enable(1); eng.train_step(xd, yd); torch.cuda.synchronize(); lib.s3d_cov_enable(0)
for k, v in sorted(C.collect(lib).items()):
    if 'fat' in k or 'gemm' in k: print(k, v)
