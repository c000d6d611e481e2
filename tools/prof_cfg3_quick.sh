# quick look at cfg3 (GPU box): the bench line's step time and the kernel trace of a few steps
export TMPDIR=/tmp
python3 bench.py --workload mul_cleanup --steps 10 --warmup 3 --no-cpu > gpurun_out/mc.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('gpurun_out/mc.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'emit ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- python3 bench.py --workload mul_cleanup --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python3 profiles/summarize_rocpd.py /tmp/p1/t_results.db | head -${1:-24}
