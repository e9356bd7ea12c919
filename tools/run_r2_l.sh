cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py -x -q -k fused_layernorm 2>&1 | grep -E "^E|assert|Error" | head -20
