# cfg5 adjacency: look-ups and table build as one instruction stream (SYMGPU_M4R_FUSE_BUILD, library built with TUNING=1) per tile height
for r in ${@:-24 40 48}; do for f in 0 1; do
  SYMGPU_M4R_R=$r SYMGPU_M4R_FUSE_BUILD=$f python3 bench.py --workload adjacency --no-cpu --no-api --steps 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$r fuse=$f  ms_per_step %.2f  kernel ms %.2f  lds frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['lds']['frac']))"
done; done
