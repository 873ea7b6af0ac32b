#!/bin/bash
# Regenerate every measured file of a round under profiles/ in ONE GPU-box session, at ONE commit.
#
#   gpurun --timeout 2400 -- "COMMIT=$(git rev-parse --short HEAD) bash tools/profile_round.sh r04"
#
# writes gpurun_out/profiles_<tag>/<tag>_* (copy them into profiles/ afterwards and commit) and
# gpurun_out/profiles_<tag>/traffic.json:
#   <tag>_provenance.json                commit, box (host, GPU id), date, ROCm -- every JSON below carries it too
#   <tag>_pytest_gpu.txt                 pytest -m gpu summary line (TESTS=0 skips)
#   <tag>_bench_line.json, <tag>_bench_default.json   python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command): its ONE stdout
#                                        line and the full record (ONLY=bench: just these, with <tag>_bench_default_provenance.json)
#   <tag>_{swe4096,rk4096,sph2048}_{kernel_stats.csv,pmc.json}   the SURVEY 8(f4) one-launch kernels (tools/pmc_leg.sh)
#   <tag>_xcdbar_probe.txt               barrier confined to one XCD vs chip-wide (tools/xcdbar_probe.hip, gridbar_probe.hip)
#   <tag>_default16384_fm{1,0}_kernel_stats.csv, _pmc.json      the headline launch, both builds
#   <tag>_sedov{8192,4096}_kernel_stats.csv, _pmc.json         north_star's target size, config 3
#   traffic.json                         per size and build: bytes / instructions per cell update (tools/make_traffic.py)
#   <tag>_adv{2048,8192}_kernel_stats.csv, <tag>_adv_pmc.json  advection, steps-per-launch kernel
#   <tag>_mg4096_kernel_stats.csv, <tag>_mg_vcycle_by_size.txt  multigrid
#   <tag>_also_traffic.json              fabric bytes of the advection launch / the V-cycle (tools/pmc_also.sh)
# Counters are collected in passes of their own (--pmc with --kernel-trace only).
TAG=${1:-r04}
R=$(pwd); O=$R/gpurun_out; P=$O/profiles_$TAG; mkdir -p $P; export TMPDIR=/tmp
GPUID=$(cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | awk '/unique_id/ && $2 != 0 {print $2; exit}')
python - > $P/${TAG}_provenance.json <<PY
import json, platform, subprocess, time
def sh(c):
    try: return subprocess.run(c, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception: return ""
print(json.dumps({"commit": "${COMMIT:-unknown}", "tag": "$TAG", "date": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
                  "box": {"host": platform.node(), "gpu_unique_id": "${GPUID:-unknown}",
                          "gpu": sh("rocminfo 2>/dev/null | grep -m1 'Marketing Name.*MI' | sed 's/.*: *//'")},
                  "rocm": sh("cat /opt/rocm/.info/version 2>/dev/null")}, indent=1))
PY
cat $P/${TAG}_provenance.json
export PYRO_PROVENANCE=$P/${TAG}_provenance.json
stats() {   # stats <name> <command...>: rocprofv3 kernel statistics of a command -> <tag>_<name>_kernel_stats.csv
  local name=$1; shift
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$name -- "$@" > $O/rocprof_${TAG}_$name.log 2>&1 )
  find $O/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $P/${TAG}_${name}_kernel_stats.csv
  head -4 $P/${TAG}_${name}_kernel_stats.csv 2>/dev/null | cut -c1-200
}
if [ "${ONLY:-}" = "bench" ]; then
  # only the driver's command again, at a later commit than the rest of the set: its own stamp
  mv $P/${TAG}_provenance.json $P/${TAG}_bench_default_provenance.json
  ( time timeout 600 python bench.py < /dev/null > $P/${TAG}_bench_line.json ) 2> $O/bench_${TAG}.err
  cp $O/bench_full.json $P/${TAG}_bench_default.json
  head -c 300 $P/${TAG}_bench_line.json; echo; tail -3 $O/bench_${TAG}.err
  exit 0
fi
if [ "${TESTS:-1}" = "1" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_${TAG}.log 2>&1
  grep -E "passed|failed|error" $O/pytest_gpu_${TAG}.log | tail -2 > $P/${TAG}_pytest_gpu.txt; cat $P/${TAG}_pytest_gpu.txt
fi
# the driver's command: the ONE compact stdout line (<tag>_bench_line.json) and the full record it
# points at (gpurun_out/bench_full.json -> <tag>_bench_default.json, the name of the earlier rounds)
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $P/${TAG}_bench_line.json ) 2> $O/bench_${TAG}.err
cp $O/bench_full.json $P/${TAG}_bench_default.json
wc -c $P/${TAG}_bench_line.json; head -c 400 $P/${TAG}_bench_line.json; echo; grep -E "^real" $O/bench_${TAG}.err
B="python $R/bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline"
for fm in 1 0; do
  stats default16384_fm$fm $B --fast-math $fm
  TRAFFIC=1 NX=16384 FM=$fm KS=-1 TAG=pmc_${TAG}_16384_fm$fm bash tools/pmc_step.sh > $O/pmc_${TAG}_16384_fm$fm.txt 2>&1
  cp $O/pmc_${TAG}_16384_fm${fm}_summary.json $P/${TAG}_default16384_fm${fm}_pmc.json
done
for nx in 8192 4096; do
  stats sedov$nx $B --nx $nx --fast-math 1
  TRAFFIC=1 NX=$nx FM=1 KS=-1 TAG=pmc_${TAG}_${nx}_fm1 bash tools/pmc_step.sh > $O/pmc_${TAG}_${nx}_fm1.txt 2>&1
  cp $O/pmc_${TAG}_${nx}_fm1_summary.json $P/${TAG}_sedov${nx}_pmc.json
done
( cd $P; python $R/tools/make_traffic.py ${TAG}_default16384_fm1_pmc.json ${TAG}_default16384_fm0_pmc.json \
    ${TAG}_sedov8192_pmc.json ${TAG}_sedov4096_pmc.json > /dev/null )
for nx in 2048 8192; do NX=$nx stats adv$nx python $R/tools/also_run.py adv; done
stats mg4096 python $R/tools/also_run.py mg
python - > $P/${TAG}_adv_pmc.json <<PY
import json, subprocess
out = {"provenance": json.load(open("$P/${TAG}_provenance.json"))}
for nx in (2048, 8192):
    subprocess.run(f"NX={nx} MULTI=3 TAG=pmcadv_${TAG}_{nx} bash tools/pmc_adv.sh > $O/pmcadv_${TAG}_{nx}.txt 2>&1", shell=True)
    out[str(nx)] = json.load(open(f"$O/pmcadv_${TAG}_{nx}_summary.json"))
print(json.dumps(out, indent=1))
PY
TAG=$TAG bash tools/pmc_also.sh > $O/pmc_also_${TAG}.log 2>&1; cp $O/${TAG}_also_traffic.json $P/ 2>/dev/null
python tools/mg_sizes.py > $P/${TAG}_mg_vcycle_by_size.txt 2>&1; cat $P/${TAG}_mg_vcycle_by_size.txt
# the SURVEY 8(f4) kernels: kernel statistics + counters per leg (tools/pmc_leg.sh)
RT=$TAG
for spec in "swe k_sw_wave 4096 swe4096" "sph k_sph_wave 2048 sph2048"; do
  set -- $spec
  LEG=$1 KN=$2 NX=$3 TAG=${RT}_$4 bash tools/pmc_leg.sh > $O/pmc_leg_${RT}_$4.log 2>&1
  cp $O/${RT}_$4_kernel_stats.csv $O/${RT}_$4_pmc.json $P/ 2>/dev/null
done
# (compressible_rk: the stages with the Runge-Kutta combination folded in -- template arguments
# <SOLVER, STD, MOL, ONE, RKF, SRC, FINT> = <0, true, true, false, true, true, 0 | 1> -- averaged over stages 1-3 of RK4:
# two middle-stage launches and the last one)
LEG=rk KN="true, true, false, true, true, " NX=4096 TAG=${RT}_rk4096 bash tools/pmc_leg.sh > $O/pmc_leg_${RT}_rk4096.log 2>&1
cp $O/${RT}_rk4096_kernel_stats.csv $O/${RT}_rk4096_pmc.json $P/ 2>/dev/null
# what a phase boundary costs: chip-wide barrier vs one confined to an XCD
( timeout 120 tools/bin/xcdbar_probe; timeout 60 tools/bin/gridbar_probe ) > $P/${TAG}_xcdbar_probe.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
ls -la $P
