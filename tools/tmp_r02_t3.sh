python -m pytest tests/test_gpu_round2.py tests/test_gpu_comm.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r02_t3_new.log
python -m pytest tests -m gpu -q --deselect tests/test_gpu_comm.py --deselect tests/test_gpu_round2.py 2>&1 | tail -15 > gpurun_out/r02_t3_old.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r02_b3_c2.json 2> gpurun_out/r02_b3_c2.err
python bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline --no-elbo-tol > gpurun_out/r02_b3_c3.json 2> gpurun_out/r02_b3_c3.err
python bench.py --config c5 --steps 20 --warmup 5 > gpurun_out/r02_b3_c5.json 2> gpurun_out/r02_b3_c5.err
AGP_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02_b3_share2_c2.json 2> gpurun_out/r02_b3_share2_c2.err
AGP_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --config c5 --steps 6 --warmup 2 > gpurun_out/r02_b3_share2_c5.json 2> gpurun_out/r02_b3_share2_c5.err
echo done
