mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_auto.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_auto.log | cut -c1-900
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --share-device --batch 16384 --steps 60 --warmup 5 > gpurun_out/bench_2proc_gloo.log 2>&1; echo "2proc rc=$?"; tail -3 gpurun_out/bench_2proc_gloo.log | cut -c1-1200
