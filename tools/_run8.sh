cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "attn or captured or pairs" 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 > gpurun_out/r02g_bench_full.json 2> gpurun_out/r02g_bench_full.err; tail -c 6000 gpurun_out/r02g_bench_full.json; tail -3 gpurun_out/r02g_bench_full.err
