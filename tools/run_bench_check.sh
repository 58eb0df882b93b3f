set -x
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/bench_n1.log 2>&1
tail -c 3000 gpurun_out/bench_n1.log | head -c 400
export GM_BENCH_BACKEND=gloo GM_BENCH_SINGLE_DEVICE=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --snark-logn 18 --psnark-logn 14 --strong-msm-logn 20 --no-cpu-baseline > gpurun_out/bench_n2.log 2>&1
g++ -O2 -std=c++17 -fPIC -shared -o /tmp/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp -ldl -lrt -lpthread
GM_BENCH_TRANSPORT=rccl GM_RCCL_LIB=/tmp/libfake_rccl.so python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --steps 5 --warmup 2 --snark-logn 18 --psnark-logn 14 --strong-msm-logn 20 --no-cpu-baseline > gpurun_out/bench_n4_fakerccl.log 2>&1
tail -n 3 gpurun_out/bench_n2.log | cut -c1-600
tail -n 3 gpurun_out/bench_n4_fakerccl.log | cut -c1-600
