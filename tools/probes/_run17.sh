mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/eigh_occ_ab.txt
for k in 0 3 2 0 3 2; do
echo "== TTR_KNOB_EIGH_BIG_OCC = $k" >> gpurun_out/r06/eigh_occ_ab.txt
TTR_KNOBS=18=$k timeout 300 python tools/decay_probe.py 0.5 4096 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r06/eigh_occ_ab.txt
done
cat gpurun_out/r06/eigh_occ_ab.txt
