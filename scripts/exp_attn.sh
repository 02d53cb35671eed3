export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_a
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_a -o attn -- python $R/scripts/${ATTN_SCRIPT:-attn_time.py} 4096 > $R/gpurun_out/prof_a.log 2>&1); echo "rc=$?"
grep -v "amdgpu.ids\|simple_timer\|output_stream" $R/gpurun_out/prof_a.log | tail -4
db=$(find $R/gpurun_out/prof_a -name "*.db" | head -1)
python $R/scripts/prof_by_grid.py $db 1 | head -14
find $R/gpurun_out/prof_a -name "*.db" -delete
