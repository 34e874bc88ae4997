export TMPDIR=/tmp
python -m pytest tests/test_train_hip.py -x -q 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -12
python3 tools/train_step_trace.py 20 2>&1 | tail -1
