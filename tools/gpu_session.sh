cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_device_compressible.py -m gpu -x -q 2>&1 | tail -2
NX=16384 bash tools/fused_ab.sh libpyrohip.so 2>&1 | tail -3
TAG=r03o PMC="1" bash tools/gpu_r03.sh 2>&1 | grep "valu_per_cell"
python -c "
import json;d=json.load(open('gpurun_out/pmc_r03o_fm1_summary.json'))
print('valu/cell', d['valu_per_cell_update'], 'busy', d['SQ_ACTIVE_INST_VALU']*4/(1024*2.4e9)*1e3/(d['GRBM_GUI_ACTIVE']/8/2.4e9*1e3), 'wait_any', d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES'], 'wait_inst', d['SQ_WAIT_INST_ANY']/d['SQ_WAVE_CYCLES'], 'lds', d['SQ_INSTS_LDS']/d['SQ_WAVES']/136)"
