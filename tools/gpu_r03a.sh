# round-3 GPU job A: resident rotation kernel — tests, timings, kernel trace
export TMPDIR=/tmp
out=gpurun_out/r03a; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rotate_resident.py -x -q -m gpu > $out/pytest_resident.log 2>&1; echo "pytest resident rc=$?" >> $out/pytest_resident.log
tail -15 $out/pytest_resident.log
ROT_TRACE=1 timeout 300 python3 tools/bench_rot.py > $out/rot_plain.out 2>&1; cat $out/rot_plain.out
ROT_ONLY_CFG2=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/rot -o t -- python3 tools/bench_rot.py > $out/rot.out 2> $out/rot.log
{ grep rotation $out/rot.out; python3 profiles/summarize_rocpd.py $out/rot/t_results.db | head -14; } > $out/rot_kernel_trace.txt
cut -c1-160 $out/rot_kernel_trace.txt
