cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graphed_gpu.py -q -m gpu -k "trains_under_replay" 2>&1 | grep -E "AssertionError|assert|passed|failed" | cut -c1-700 | head -12
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "fusion_backward_cut" 2>&1 | grep -E "Error|assert|passed|failed|pipeline.py|med.py|fusion_ops.py" | cut -c1-900 | head -30
