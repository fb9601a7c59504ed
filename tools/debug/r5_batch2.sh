cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/r5_clock2.sh > gpurun_out/r5clock/clock2.log 2>&1
timeout 900 python -m pytest tests/test_gpu_groupwise.py tests/test_gpu_workspace.py tests/test_gpu_tiled.py tests/test_gpu_kv_fp8.py tests/test_gpu_attn_prefill.py tests/test_gpu_toy_pipeline.py tests/test_gpu_prefill_objects.py -x -q -m gpu > gpurun_out/r5_tests1.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5_tests1.log
timeout 600 python bench.py > gpurun_out/r5_bench_base.json 2> gpurun_out/r5_bench_base.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r5_bench_base.json
bash tools/bench_ab.sh "main t13" 2 "--workload qwenvl_7b" > gpurun_out/r5_qwenvl_ab.txt 2>&1; cat gpurun_out/r5_qwenvl_ab.txt
