"""CPU: the timm-0.3.2 restatement (oracle/timm_shim + oracle.voxel_oracle block functions) has no
reference-side golden (timm is un-vendored; SURVEY.md section 8(c): "parity unpinned").  Cross-check it against
independent implementations of the same published algorithm."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from oracle import voxel_oracle as vo

SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'timm_shim')


def _shim_block(D, H):
    sys.path.insert(0, SHIM)
    try:
        from timm.models.vision_transformer import Block
    finally:
        sys.path.remove(SHIM)
    from functools import partial
    return Block(D, H, mlp_ratio=4., qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))


@pytest.mark.parametrize('D,H,N', [(384, 6, 26), (768, 3, 15), (192, 3, 257)])
def test_block_vs_independent_sdpa(D, H, N):
    torch.manual_seed(1)
    blk = _shim_block(D, H).eval()
    for p in blk.parameters():
        torch.nn.init.normal_(p, 0, 0.05)
    x = torch.randn(3, N, D)
    sd = {'blocks.0.' + k: v for k, v in blk.state_dict().items()}
    y_shim = blk(x)
    y_fn = vo.vit_block(x, sd, 0, H)
    # independent: F.scaled_dot_product_attention + manual residuals
    p = 'blocks.0.'
    h = F.layer_norm(x, (D,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-6)
    qkv = F.linear(h, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'])
    q, k, v = [t.reshape(3, N, H, D // H).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(3, N, D)
    x1 = x + F.linear(a, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'])
    h2 = F.layer_norm(x1, (D,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-6)
    y_ind = x1 + F.linear(F.gelu(F.linear(h2, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'])),
                          sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
    assert float((y_shim - y_ind).abs().max()) < 2e-5
    assert float((y_fn - y_ind).abs().max()) < 2e-5


def test_attention_vs_nn_multihead_attention():
    torch.manual_seed(2)
    D, H, N = 192, 3, 17
    mha = torch.nn.MultiheadAttention(D, H, batch_first=True).eval()
    x = torch.randn(2, N, D)
    sd = {'a.qkv.weight': mha.in_proj_weight.detach(), 'a.qkv.bias': mha.in_proj_bias.detach(),
          'a.proj.weight': mha.out_proj.weight.detach(), 'a.proj.bias': mha.out_proj.bias.detach()}
    ref, _ = mha(x, x, x, need_weights=False)
    got = vo.attention(x, sd, 'a.', H)
    assert float((ref - got).abs().max()) < 1e-5


def test_group_encoder_layer_vs_torch_module():
    """oracle.group_encoder_layer == nn.TransformerEncoderLayer(d_model=D, dim_feedforward=D, nhead=4) in eval,
    fed seq-first exactly like vit_3d_2d_pretrain.py:479."""
    torch.manual_seed(3)
    D = 64
    layer = torch.nn.TransformerEncoderLayer(d_model=D, dim_feedforward=D, nhead=4).eval()
    sd = {'group_embed.' + k: v.detach() for k, v in layer.state_dict().items()}
    x = torch.randn(18, 4, D)
    ref = layer(x)
    got = vo.group_encoder_layer(x, sd)
    assert float((ref - got).abs().max()) < 1e-5


def test_transformers_vit_layer_agrees():
    tr = pytest.importorskip('transformers')
    try:
        from transformers import ViTConfig
        from transformers.models.vit.modeling_vit import ViTLayer
    except Exception:
        pytest.skip('transformers ViT layer unavailable')
    torch.manual_seed(4)
    D, H, N = 192, 3, 10
    cfg = ViTConfig(hidden_size=D, num_attention_heads=H, intermediate_size=4 * D, hidden_act='gelu',
                    layer_norm_eps=1e-6, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    try:
        cfg._attn_implementation = 'eager'
        layer = ViTLayer(cfg).eval()
    except Exception:
        pytest.skip('cannot build ViTLayer')
    hf = layer.state_dict()

    def pick(*cands):
        for c in cands:
            if c in hf:
                return hf[c]
        raise KeyError(cands)
    try:
        qw = torch.cat([pick('attention.attention.query.weight'), pick('attention.attention.key.weight'),
                        pick('attention.attention.value.weight')])
        qb = torch.cat([pick('attention.attention.query.bias'), pick('attention.attention.key.bias'),
                        pick('attention.attention.value.bias')])
        sd = {'blocks.0.norm1.weight': pick('layernorm_before.weight'), 'blocks.0.norm1.bias': pick('layernorm_before.bias'),
              'blocks.0.attn.qkv.weight': qw, 'blocks.0.attn.qkv.bias': qb,
              'blocks.0.attn.proj.weight': pick('attention.output.dense.weight'),
              'blocks.0.attn.proj.bias': pick('attention.output.dense.bias'),
              'blocks.0.norm2.weight': pick('layernorm_after.weight'), 'blocks.0.norm2.bias': pick('layernorm_after.bias'),
              'blocks.0.mlp.fc1.weight': pick('intermediate.dense.weight'), 'blocks.0.mlp.fc1.bias': pick('intermediate.dense.bias'),
              'blocks.0.mlp.fc2.weight': pick('output.dense.weight'), 'blocks.0.mlp.fc2.bias': pick('output.dense.bias')}
    except KeyError:
        pytest.skip('unexpected transformers ViTLayer parameter names')
    x = torch.randn(2, N, D)
    with torch.no_grad():
        out = layer(x)
        ref = out[0] if isinstance(out, (tuple, list)) else out
        got = vo.vit_block(x, sd, 0, H)
    assert float((ref - got).abs().max()) < 2e-5
