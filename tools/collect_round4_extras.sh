# Round-4 extras kept under profiles/ (run on the GPU box through gpurun):
#   r04_chain_constants.txt                  tools/ubench/lat (dependent-issue latencies) + dpp16 (single-wave DPP elimination)
#   r04_tile_factorisation_decomposition.txt tools/bench_diag.py
#   r04_soaks.txt                            determinism soaks of the final build, incl. the split launches (C3 shape, 8 latents)
#   r04_c2_bench_line.json                   the default bench line with the CPU oracle RUN through all three ELBO rules
export TMPDIR=/tmp
O=gpurun_out
{ echo "# tools/ubench/lat (hipcc --offload-arch=gfx950 -O3): single-wave chains of 64 unrolled operations"; (cd tools/ubench && ./lat);
  echo; echo "# tools/ubench/dpp16: 16x16 [A | I] Gauss-Jordan in ONE wave, multipliers by DPP row_newbcast"; (cd tools/ubench && ./dpp16); } > $O/r04_chain_constants.txt 2>&1
timeout 600 python tools/bench_diag.py > $O/r04_tile_factorisation_decomposition.txt 2>&1
{ echo "# final build of round 4 (C3 shape and the 8-latent batch run as SPLIT launches: chain kernel + tile kernel)";
  echo "## C2 shape fp64, 2 x 60000 steps (merged launch, look-ahead on)"; timeout 900 python tools/soak_determinism.py 60000;
  echo "## C2 shape fp64, split launch forced (AGP_CHAIN_SPLIT=1), 2 x 20000 steps"; AGP_CHAIN_SPLIT=1 timeout 900 python tools/soak_determinism.py 20000;
  echo "## C3 shape fp32 m = B = 2048, 2 x 20000 steps (split launch)"; timeout 900 python tools/soak_determinism.py 20000 2048 f32;
  echo "## 8 latents (C4 shape: split launch of 8 chains + tiles, two look-ahead streams), 2 x 5000 steps"; timeout 900 python tools/soak_multilatent.py 8 5000 2;
  echo "## 5 latents, 2 x 3000 steps"; timeout 900 python tools/soak_multilatent.py 5 3000 2; } 2>&1 | grep -v amdgpu.ids > $O/r04_soaks.txt
timeout 900 python bench.py --cpu-elbo-seconds 320 > $O/r04_c2_bench_line_full.json 2> $O/r04_c2_full.err
tail -3 $O/r04_soaks.txt; tail -5 $O/r04_chain_constants.txt; tail -c 1500 $O/r04_c2_bench_line_full.json
