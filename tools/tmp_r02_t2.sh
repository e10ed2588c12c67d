set -x
python -m pytest tests/test_gpu_round2.py tests/test_gpu_comm.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r02_t2_new.log
python -m pytest tests -m gpu -q --deselect tests/test_gpu_comm.py --deselect tests/test_gpu_round2.py 2>&1 | tail -15 > gpurun_out/r02_t2_old.log
python bench.py --steps 100 --warmup 20 > gpurun_out/r02_b_c2.json 2> gpurun_out/r02_b_c2.err
python bench.py --config c3 --steps 50 --warmup 10 > gpurun_out/r02_b_c3.json 2> gpurun_out/r02_b_c3.err
python bench.py --config c4 --steps 50 --warmup 10 > gpurun_out/r02_b_c4.json 2> gpurun_out/r02_b_c4.err
python bench.py --config c5 --steps 20 --warmup 5 > gpurun_out/r02_b_c5.json 2> gpurun_out/r02_b_c5.err
AGP_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02_b_share2_c2.json 2> gpurun_out/r02_b_share2_c2.err
AGP_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config c4 --hyper-every 5 --steps 20 --warmup 5 > gpurun_out/r02_b_share2_c4.json 2> gpurun_out/r02_b_share2_c4.err
tail -5 gpurun_out/r02_t2_new.log gpurun_out/r02_t2_old.log
tail -c 600 gpurun_out/r02_b_*.json
tail -3 gpurun_out/r02_b_*.err
