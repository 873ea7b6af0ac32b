cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_device_advection.py -m gpu -x -q 2>&1 | tail -2
SIZES="2048:13,12,13;8192:48,32,48" timeout 300 python tools/adv_time.py 2>&1
