set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
IFX_PAIR_FORWARDS=0 python -m pytest tests/test_hip_model.py tests/test_hip_sequence_parallel.py tests/test_hip_plugin_api.py tests/test_hip_quant.py -q -m gpu -x > $OUT/r5m_nopair.log 2>&1; echo "nopair rc=$?" >> $OUT/r5m_nopair.log
IFX_PAIR_MODE=lockstep python -m pytest tests/test_hip_model.py tests/test_hip_sequence_parallel.py tests/test_hip_plugin_api.py tests/test_hip_quant.py -q -m gpu -x > $OUT/r5m_lockstep.log 2>&1; echo "lockstep rc=$?" >> $OUT/r5m_lockstep.log
IFX_V_DIRECT=0 python -m pytest tests/test_hip_model.py -q -m gpu -x -k "full_size" > $OUT/r5m_novd.log 2>&1; echo "novd rc=$?" >> $OUT/r5m_novd.log
tail -n 3 $OUT/r5m_nopair.log $OUT/r5m_lockstep.log $OUT/r5m_novd.log
