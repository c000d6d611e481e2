# A/B of the XCD-aware orders on this box (GPU box; needs variants/libsymgpu_v{0,1,2}.so: neither / output stage / output stage + row stream)
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -4
cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>/dev/null | head -4
for rep in 1 2; do
  for v in v0 v1; do cp variants/libsymgpu_$v.so symmer_amd/libsymgpu.so; python bench.py --workload mul_cleanup --no-extras --no-cpu --no-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v emit', round(d['roofline']['avg_launch_ms'],4))"; done
  for v in v1 v2; do cp variants/libsymgpu_$v.so symmer_amd/libsymgpu.so; python bench.py --workload product --steps 3 --warmup 1 --no-extras --no-cpu --no-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rows', round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],1))"; done
done
