"""TEST INFRASTRUCTURE ONLY -- a minimal restatement of the timm==0.3.2 API surface
that the reference touches (requirements.txt:6, README.md:22 pin timm 0.3.2, which is
NOT vendored under /root/reference and cannot be installed here).

Only oracle/, tests/ and the golden-vector generator may put this directory on sys.path.
The product package (simple3d-former_amd/) never imports it.

PARITY NOTE: the reference holds no tests or golden vectors for the timm boundary, so
this restatement is "parity unpinned" by the reference itself.  It is cross-checked in
tests/test_oracle_timm.py against independent implementations of the same published
algorithm (torch.nn.functional.scaled_dot_product_attention, nn.MultiheadAttention and
transformers' ViT layer).
"""
__version__ = "0.3.2-shim"
