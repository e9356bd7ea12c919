"""simple3d-former_amd: MI355X-native (gfx950) implementation of the Simple3D-Former training hot path.

Public names mirror the reference's module API (models/embed_layer_3d_modality.py, models/vit_3d_2d_pretrain.py and
the timm==0.3.2 VisionTransformer it subclasses); everything computes through libs3d_hip.so (include/s3d_hip.h)."""
from . import _lib, binvox, metrics
from .engine import BACKBONES, VoxelEngine, voxel_param_shapes
from .tokenizers import VoxelEmbed, VoxelEmbed_no_average, VoxelNaiveProjection
from .metrics import ClsEvaluator, PartSegEvaluator
from .point_engine import PointEngine, point_param_shapes
from .point_model import (PointTransformerCls, PointTransformerSeg, PointTransformerSeg0Layer, PointTransformerSeg1Layer,
                          PointTransformerSegLWF, model_module)
from .voxel_model import AMSoftmaxLayer, Attention, Block, Feature3D_ViT2D_V2, Mlp, PatchEmbed, VisionTransformer

__all__ = ['VoxelEmbed', 'VoxelEmbed_no_average', 'VoxelNaiveProjection', 'VisionTransformer', 'Block', 'Attention',
           'Mlp', 'PatchEmbed', 'AMSoftmaxLayer', 'Feature3D_ViT2D_V2', 'VoxelEngine', 'BACKBONES',
           'voxel_param_shapes', 'PointEngine', 'point_param_shapes', 'PointTransformerCls', 'PointTransformerSeg', 'PointTransformerSeg1Layer',
           'PointTransformerSeg0Layer', 'PointTransformerSegLWF', 'model_module', 'ClsEvaluator',
           'PartSegEvaluator', 'binvox', 'metrics']
