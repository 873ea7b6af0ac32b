cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_device_multigrid.py -m gpu -x -q 2>&1 | tail -2
for lib in libv_np.so libpyrohip.so libv_np.so libpyrohip.so; do echo $lib; PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/$lib MG_SIZES=2048,4096 timeout 300 python tools/mg_sizes.py; done
