cd $GRAFT_REPO_ROOT
for lib in libv_np.so libpyrohip.so; do echo $lib; PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/$lib SIZES="2048:13,13,12;8192:48,48" timeout 300 python tools/adv_time.py 2>&1; done
STEPS=12 SIZES="2048:0;4096:0;8192:0;16384:0" timeout 600 python tools/march_sweep.py
FM=0 STEPS=12 SIZES="4096:0;16384:0" timeout 600 python tools/march_sweep.py
