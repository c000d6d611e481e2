export TMPDIR=/tmp
out=gpurun_out/r03h; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_rotate_resident.py tests/test_gpu_parity.py -x -q -m gpu -k "chain or rotat or circuit or symmerlator" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 300 python3 tools/bench_chain3.py 2>&1
timeout 300 python3 tools/bench_chain5.py 2>&1
