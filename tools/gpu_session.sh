cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_device_advection.py tests/test_zz_comm.py tests/test_host_api.py -m gpu -x -q 2>&1 | tail -2
BOTH=1 SIZES="2048:0,13,13;8192:0,48,32" timeout 300 python tools/adv_time.py 2>&1
