cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -2 gpurun_out/bench_final.err
