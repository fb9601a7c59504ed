#!/bin/bash
# repeat the single-rank torchrun bench (tests/test_gpu_multi.py) N times; keep the full stderr of failing runs
N=${1:-20}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/trl
for i in $(seq 1 $N); do
  HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29600 + i)) \
    bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-full-prefill > gpurun_out/trl/out_$i.txt 2> gpurun_out/trl/err_$i.txt
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -eq 0 ]; then rm -f gpurun_out/trl/out_$i.txt gpurun_out/trl/err_$i.txt; fi
done
