#!/bin/bash
# Profiles of one round, collected on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# -> gpurun_out/<tag>/: kernel trace of bench.py (single stream), separate --pmc passes (HBM bytes, MFMA busy),
#    the MFMA-busy calibration, pmc_latest.json and readable summaries.  Copy what should be judged into profiles/.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=2048
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --single-stream --no-cpu-baseline --no-extras > $OUT/kt_bench.json 2> $OUT/kt.err
python $R/tools/rocprof_summary.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) > $OUT/kernel_stats.txt 2>> $OUT/kt.err
for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python $R/tools/pmc_step.py $B > $OUT/pmc_$name.log 2>&1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $R/tools/microbench.hip -o /tmp/ttr_microbench 2> /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_calib -o p -- /tmp/ttr_microbench m > $OUT/pmc_mfma_calib.log 2>&1
python $R/tools/pmc_to_json.py $OUT 2 $B > $OUT/pmc_summary.txt 2> $OUT/pmc_summary.err
tail -n 30 $OUT/pmc_summary.txt
head -n 25 $OUT/kernel_stats.txt
# ---- the inputs / configs furthest from their roofs ("Missing 3"): the decaying-spectrum variant of the metric
#      (no shortcut fires), BASELINE C3's per-GPU share and C4 -- kernel trace + the same three counter passes each
for spec in "decay05:2048:decay0.5:2" "decay10:2048:decay1.0:2" "c3:64:c3:2" "c4:1:c4:2"; do
  IFS=: read name bb kind st <<< "$spec"
  D=$OUT/$name
  mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -o kt -- python $R/tools/pmc_step.py $bb $kind $st > $D/kt.log 2> $D/kt.err
  python $R/tools/rocprof_summary.py $(find $D/kt -name "*kernel_trace.csv" | head -1) > $D/kernel_stats.txt 2>> $D/kt.err
  for cs in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    pn=${cs%%:*}; ctrs=${cs#*:}
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $D/pmc_$pn -o p -- python $R/tools/pmc_step.py $bb $kind $st > $D/pmc_$pn.log 2>&1
  done
  cp -r $OUT/pmc_mfma_calib $D/pmc_mfma_calib 2> /dev/null
  python $R/tools/pmc_to_json.py $D $st $bb > $D/pmc_summary.txt 2> $D/pmc_summary.err
  head -n 14 $D/pmc_summary.txt
  head -n 14 $D/kernel_stats.txt
  rm -rf $D/kt $D/pmc_fetch $D/pmc_write $D/pmc_mfma $D/pmc_mfma_calib
done
# ---- fp64: a resident batch of BASELINE config C2 (256 rank-64 trains, 10 cores x mode 128), same views
F=$OUT/fp64
mkdir -p $F
rocprofv3 --kernel-trace --stats --output-format csv -d $F/kt -o kt -- python $R/tools/c2_step.py 256 3 > $F/kt.log 2> $F/kt.err
python $R/tools/rocprof_summary.py $(find $F/kt -name "*kernel_trace.csv" | head -1) > $F/kernel_stats.txt 2>> $F/kt.err
for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $F/pmc_$name -o p -- python $R/tools/c2_step.py 256 2 > $F/pmc_$name.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $F/pmc_mfma_calib -o p -- /tmp/ttr_microbench d > $F/pmc_mfma_calib.log 2>&1
python $R/tools/pmc_to_json.py $F 2 256 > $F/pmc_summary.txt 2> $F/pmc_summary.err
tail -n 24 $F/pmc_summary.txt
head -n 22 $F/kernel_stats.txt

