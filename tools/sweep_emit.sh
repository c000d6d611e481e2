# emit-stage shape sweep at cfg3 (GPU box): SYMGPU_EMIT_SHAPE = "bitmap words per wavefront, 64-chunk steps in flight"
for rep in 1 2; do
for shape in 4,4 2,4 1,4 1,8 2,2 2,8 1,2; do
  SYMGPU_EMIT_SHAPE=$shape python3 bench.py --workload mul_cleanup --steps 10 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$shape', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
done
SYMGPU_EMIT_NO_ONE_OUTER=1 SYMGPU_EMIT_SHAPE=2,4 python3 bench.py --workload mul_cleanup --steps 10 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2,4 two-gather', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
done
