mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/gputest_full.txt 2>&1
tail -4 gpurun_out/r06/gputest_full.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_full_run1.json 2> gpurun_out/r06/bench_full_run1.err
cp profiles/bench_full_latest.json gpurun_out/r06/bench_full_run1_full.json
tail -c 3000 gpurun_out/r06/bench_full_run1.json
