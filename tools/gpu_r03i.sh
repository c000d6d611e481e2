export TMPDIR=/tmp
out=gpurun_out/r03i; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_rotate_resident.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz2.py -x -q -m gpu -k "rotat or resident or chain" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
ROT_TRACE=1 timeout 300 python3 tools/bench_rot.py 2>&1 | cut -c1-260
timeout 300 python3 tools/bench_perform.py 2>&1 | head -3
