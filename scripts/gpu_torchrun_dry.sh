cd /root/repo
# dry run of the N>1 control flow on one GPU: 2 ranks share cuda:0, collectives over gloo
MOJO_BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/torchrun_dry.log 2>&1
echo rc=$?; tail -3 gpurun_out/torchrun_dry.log | cut -c1-1500
