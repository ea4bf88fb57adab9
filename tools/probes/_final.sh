bash tools/profile_round.sh r06 > gpurun_out/r06_profile.log 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/r06/pmc_latest.json profiles/pmc_latest.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err
cp profiles/bench_full_latest.json gpurun_out/r06/bench_final_full.json
python tools/latency_probe.py > gpurun_out/r06/latency.txt 2>&1
tail -c 600 gpurun_out/r06/bench_final.json
