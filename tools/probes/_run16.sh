mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/l1idle_ab.txt
for k in 1 3 1 3; do
TTR_KNOBS=17=$k timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs 2> /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('knob17=$k', d['ms_per_step'], d['kernel_ms_per_step']['qr_apply'], d['parity']['rel_err_vs_oracle_svd'])" >> gpurun_out/r06/l1idle_ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r06/gputest16.txt 2>&1
tail -3 gpurun_out/r06/gputest16.txt
cat gpurun_out/r06/l1idle_ab.txt
