python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
python bench.py --config c5 --no-extras --no-elbo-tol --steps 20 --warmup 5 > gpurun_out/g1_c5.json 2>gpurun_out/g1_c5.err < /dev/null
tail -c 2500 gpurun_out/g1_c5.json
