cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3
bash tools/bench_ab.sh "main ppnopf" 4 | tee gpurun_out/r5_bench_ab_pp_next_tile_prefetch.txt
