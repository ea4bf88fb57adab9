mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/pipe_ab.txt
for k in 1 2 4 1 2 4; do
TTR_KNOBS=19=$k timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs 2> /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nper=$k', d['ms_per_step'], d['kernel_ms_per_step']['qr_apply'], d['parity'])" >> gpurun_out/r06/pipe_ab.txt
done
cat gpurun_out/r06/pipe_ab.txt
TTR_KNOBS=19=4 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r06/gputest19.txt 2>&1
tail -3 gpurun_out/r06/gputest19.txt
