# round-3 GPU job F: allocator arena, sharded product + cleanup, whole GPU suite, default bench
export TMPDIR=/tmp
out=gpurun_out/r03f; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_all.log 2>&1; echo "rc=$?" >> $out/pytest_all.log; tail -4 $out/pytest_all.log
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; head -c 1800 $out/bench_default.json; echo; tail -3 $out/bench_default.err
