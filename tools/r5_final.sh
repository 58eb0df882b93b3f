#!/bin/bash
# GPU box: the measurement batch of round 5 on the library as committed (see profiles/README.md for what lands where)
O=gpurun_out
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > $O/r5_gpu_suite.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/r5_bench_driver_args.json 2> $O/r5_bench_driver_args.err
python bench.py > $O/r5_bench.json 2> $O/r5_bench.err
bash tools/profile_round.sh r5 > $O/r5_profile_round.log 2>&1
bash tools/exposed_round.sh r5 > $O/r5_exposed_round.log 2>&1
TAG=r5 bash tools/prover_sweep.sh $O/r5_prover_sweep.txt > /dev/null 2>&1
bash tools/shard_shares.sh $O/r5_shard_shares.txt > /dev/null 2>&1
python tools/footprint_table.py > $O/r5_footprint.txt 2> $O/r5_footprint.err
python tools/collective_latency.py > $O/r5_collective_latency_world1.json 2>/dev/null
for T in shm hook; do
  GM_BENCH_BACKEND=gloo GM_BENCH_SINGLE_DEVICE=1 GM_BENCH_TRANSPORT=$T timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
    bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/r5_bench_n2_shared_gpu_$T.json 2> $O/r5_bench_n2_$T.err
done
python tools/msm_sizes.py > $O/r5_msm_sizes.txt 2>/dev/null
tail -3 $O/r5_gpu_suite.txt; tail -c 600 $O/r5_bench_driver_args.json; cat $O/r5_prover_sweep.txt $O/r5_shard_shares.txt $O/r5_footprint.txt
