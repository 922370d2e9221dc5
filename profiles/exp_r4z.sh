mkdir -p gpurun_out/r4z
JMHIP_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r4z/bench_n2.json 2> gpurun_out/r4z/bench_n2.err
tail -c 1500 gpurun_out/r4z/bench_n2.json; tail -3 gpurun_out/r4z/bench_n2.err
