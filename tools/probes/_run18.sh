mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "flat_spectrum or eps or orth_fixup_split or free_function or rank" > gpurun_out/r06/gputest18.txt 2>&1
tail -3 gpurun_out/r06/gputest18.txt
python tools/latency_probe.py > gpurun_out/r06/latency2.txt 2>&1
grep "non-batch\|C2 single\|B=1:" gpurun_out/r06/latency2.txt | head -8
