#!/bin/bash
# round 4, session a: the RCCL exchange behind the C-ABI (tests + in-process bench on one device, 2 shards)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sharded_map.py -x -q 2>&1 | tail -15 > gpurun_out/r4a_sharded_tests.txt
cat gpurun_out/r4a_sharded_tests.txt
timeout 300 python bench.py --in-process --gpus 1 --shards-per-gpu 2 --rows 10000000 --steps 20 --warmup 5 --full-json gpurun_out/r4a_inproc_full.json > gpurun_out/r4a_inproc.json 2> gpurun_out/r4a_inproc.err
tail -c 3000 gpurun_out/r4a_inproc.json
RXGPU_SHARD_MERGE=host timeout 300 python bench.py --in-process --gpus 1 --shards-per-gpu 2 --rows 10000000 --steps 20 --warmup 5 --full-json gpurun_out/r4a_inproc_host_full.json > gpurun_out/r4a_inproc_host.json 2> gpurun_out/r4a_inproc_host.err
tail -c 600 gpurun_out/r4a_inproc_host.json
tail -5 gpurun_out/r4a_inproc.err
