export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_g
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_g -o bench -- python $R/bench.py --steps 17 --warmup 3 --cpu-frames 0 --no-secondary --no-roofline $SEQ_FLAGS > $R/gpurun_out/prof_g.log 2>&1); echo "rc=$?"
db=$(find $R/gpurun_out/prof_g -name "*.db" | head -1)
python $R/scripts/prof_sequence.py $db > $R/gpurun_out/sequence.txt
find $R/gpurun_out/prof_g -name "*.db" -delete
tail -3 $R/gpurun_out/sequence.txt
