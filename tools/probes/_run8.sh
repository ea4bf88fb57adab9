mkdir -p gpurun_out/r06
for c in 2 3 4 2 3; do
TTR_STREAM_CHUNKS=$c python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs 2> /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks $c', d['ms_per_step'], d['value'])" >> gpurun_out/r06/chunks_ab.txt
done
cat gpurun_out/r06/chunks_ab.txt
