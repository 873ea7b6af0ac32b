mkdir -p gpurun_out
{
python -m pytest tests/test_device_multigrid.py -m gpu -x -q -k "march or 4096 or multi_tile" 2>&1 | tail -3
for side in 1.0 1.17 1.3 1.5 1.8; do
  echo "== dirichlet SIDE=$side"
  PYRO_MG_MARCH_SIDE=$side python tools/mg_prof.py 2048 4096 2>&1 | grep -E "nx=|march"
done
} > gpurun_out/march_ab.txt 2>&1
