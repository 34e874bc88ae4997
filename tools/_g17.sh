export TMPDIR=/tmp
python -m pytest tests/test_train_hip.py -x -q 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -12
python -m pytest tests/test_hip_parity.py -x -q -k "knn or point_stage or training_step or per_frame" 2>&1 | tail -2
python3 tools/train_step_trace.py 20 2>&1 | tail -1
OCCNERF_LINEAR_RESIDENT=2 python3 tools/train_step_trace.py 20 2>&1 | tail -1
