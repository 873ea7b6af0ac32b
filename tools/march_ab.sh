mkdir -p gpurun_out
{
python -m pytest tests/test_device_multigrid.py tests/test_incompressible.py -m gpu -x -q 2>&1 | tail -3
for e in 1 0; do
  echo "== EAGER_R=$e"
  if [ $e = 1 ]; then export PYRO_MG_EAGER_R=1; else unset PYRO_MG_EAGER_R; fi
  python tools/mg_prof.py 512 2048 4096 2>&1
done
} > gpurun_out/march_ab.txt 2>&1
