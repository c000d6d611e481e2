# round-3 GPU job C: the new bench workloads, quick look (no profiler)
export TMPDIR=/tmp
out=gpurun_out/r03c; rm -rf $out; mkdir -p $out
for wl in rotation mul_cleanup gf2; do
  timeout 600 python3 bench.py --workload $wl --steps 5 --warmup 2 > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "== $wl rc=$?"; tail -c 3000 $out/bench_$wl.json; tail -3 $out/bench_$wl.err
done
timeout 900 python -m pytest tests/test_gpu_rotate_resident.py -x -q -m gpu > $out/pytest_resident.log 2>&1; tail -3 $out/pytest_resident.log
