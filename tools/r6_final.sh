#!/bin/bash
# GPU box: the measurement batch of round 6 on the library as committed (profiles/README.md says what lands where).
# Usage: bash tools/r6_final.sh [part ...]   parts: suite bench profile exposed sweep shares misc n2 soak   (default: all but soak)
O=gpurun_out
PARTS=${@:-suite bench profile exposed sweep shares misc n2}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has suite; then (time python -m pytest tests -q -m gpu --durations=30 2>&1 | tail -40) > $O/r6_gpu_suite.txt 2>&1; fi
if has bench; then
  python bench.py --steps 20 --warmup 5 > $O/r6_bench_driver_args.json 2> $O/r6_bench_driver_args.err
  python bench.py > $O/r6_bench.json 2> $O/r6_bench.err
fi
if has profile; then bash tools/profile_round.sh r6 > $O/r6_profile_round.log 2>&1; fi
if has exposed; then bash tools/exposed_round.sh r6 > $O/r6_exposed_round.log 2>&1; fi
if has sweep; then TAG=r6 bash tools/prover_sweep.sh $O/r6_prover_sweep.txt > /dev/null 2>&1; fi
if has shares; then
  bash tools/shard_shares.sh $O/r6_shard_shares.txt > /dev/null 2>&1
  bash tools/psnark_shard_shares.sh 26 $O/r6_psnark_shard_shares.txt 2> $O/r6_psnark_shard_shares.err > /dev/null
fi
if has misc; then
  python tools/footprint_table.py > $O/r6_footprint.txt 2> $O/r6_footprint.err
  python tools/msm_sizes.py > $O/r6_msm_sizes.txt 2>/dev/null
fi
if has n2; then
  # the N > 1 bench on the shared GPU (timings mean nothing there: what is recorded is that every leg runs, the routes, the proof hashes)
  for T in shm hook; do
    GM_BENCH_BACKEND=gloo GM_BENCH_SINGLE_DEVICE=1 GM_BENCH_TRANSPORT=$T timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
      bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --snark-logn 20 --psnark-logn 16 --strong-msm-logn 22 > $O/r6_bench_n2_shared_gpu_$T.json 2> $O/r6_bench_n2_$T.err
  done
  g++ -O2 -std=c++17 -fPIC -shared -o /tmp/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp -ldl -lrt -lpthread
  GM_BENCH_BACKEND=gloo GM_BENCH_SINGLE_DEVICE=1 GM_BENCH_TRANSPORT=rccl GM_RCCL_LIB=/tmp/libfake_rccl.so timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29578 \
    bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --snark-logn 18 --psnark-logn 14 --strong-msm-logn 20 > $O/r6_bench_n8_shared_gpu_fake_rccl.json 2> $O/r6_bench_n8_fake_rccl.err
fi
if has soak; then
  SOAK_SECONDS=${SOAK_SECONDS:-420} SOAK_ONLY=psnark python -m pytest tests/soak_dist_native.py -q -s 2>&1 | tail -3 > $O/r6_soak_dist_native_psnark.txt; cp $O/soak_dist_native.json $O/r6_soak_dist_native_psnark.json
  SOAK_SECONDS=${SOAK_SECONDS:-420} python -m pytest tests/soak_dist_native.py -q -s 2>&1 | tail -3 > $O/r6_soak_dist_native.txt; cp $O/soak_dist_native.json $O/r6_soak_dist_native.json
fi
tail -3 $O/r6_gpu_suite.txt 2>/dev/null; tail -c 400 $O/r6_bench_driver_args.json 2>/dev/null; cat $O/r6_prover_sweep.txt $O/r6_shard_shares.txt 2>/dev/null
