# cfg5 adjacency: one 8-bit table per step (SYMGPU_M4R7=0) against two 7-bit tables folded with XOR3 (1); library built with TUNING=1
for r in ${@:-48 40}; do for f in 0 1; do
  SYMGPU_M4R_R=$r SYMGPU_M4R7=$f python3 bench.py --workload adjacency --no-cpu --no-api --steps 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$r seven=$f  ms_per_step %.2f  kernel ms %.2f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
