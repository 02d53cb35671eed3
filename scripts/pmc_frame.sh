#!/bin/bash
# In-frame counters: rocprofv3 --pmc over the EAGER bench frame (bench.py --no-graph), one pass per
# counter group (SQ; FETCH_SIZE; WRITE_SIZE — the TCC slots do not hold both), kernel trace only; aggregated per kernel
# name over that kernel's in-frame launches into gpurun_out/pmc_frame_<mode>.json by scripts/pmc_frame.py.
#   MODE=f32 (split fp32, default) | fp16 | bf16      STEPS=frames per pass (default 6)
#   "collected_at" / "lib_digest" of the JSON = the build stamp compiled into the library that ran (sgam_build_commit / sgam_build_digest)
export TMPDIR=/tmp
MODE=${MODE:-f32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcf_$MODE
rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-6} --warmup 2 --cpu-frames 0 --no-secondary --no-roofline --no-graph --dtype $MODE > $OUT/$name.log 2>&1; echo "$MODE $name rc=$?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
STAMP=$(cd $GRAFT_REPO_ROOT && python3 -c "from sgam_neurips22_amd import _lib; l = _lib.load(); print(l.sgam_build_commit().decode(), l.sgam_build_digest().decode())" 2>/dev/null | tail -1)
python3 $GRAFT_REPO_ROOT/scripts/pmc_frame.py $OUT $GRAFT_REPO_ROOT/gpurun_out/pmc_frame_$MODE.json $((${STEPS:-6} + 2)) "library @ ${STAMP% *}" "${STAMP#* }"
find $OUT -name "*.csv" -size +8M -delete
