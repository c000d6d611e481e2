export TMPDIR=/tmp
out=gpurun_out/r03j; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_all.log 2>&1; echo "rc=$?" >> $out/pytest_all.log; grep -E "passed|failed|rc=" $out/pytest_all.log | tail -3
timeout 300 python3 tools/bench_perform.py 2>&1
