set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py tests/test_hip_sequence_parallel.py tests/test_hip_quant.py tests/test_hip_plugin_api.py tests/test_pipeline_host_api.py tests/test_hip_full_size_properties.py -q -m gpu -x > $OUT/r5j_tests.log 2>&1
echo "rc=$?" >> $OUT/r5j_tests.log
tail -n 6 $OUT/r5j_tests.log
