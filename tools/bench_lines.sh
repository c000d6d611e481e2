# GPU box: the five bench lines as the driver would see them (un-profiled; their `traffic` comes from profiles/<tag>_*traffic.json)
tag=${1:-r04}
dst=gpurun_out/${tag}_lines; rm -rf $dst; mkdir -p $dst
timeout 900 python3 bench.py > $dst/${tag}_bench_n1.json 2> $dst/bench_n1.err
for wl in rotation mul_cleanup gf2 adjacency; do
  timeout 900 python3 bench.py --workload $wl > $dst/${tag}_${wl}_n1.json 2> $dst/${wl}_n1.err
done
ls -la $dst
