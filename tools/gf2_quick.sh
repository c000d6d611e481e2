# quick look at cfg4 (GPU box): step time and per-kernel averages
export TMPDIR=/tmp
python3 bench.py --workload gf2 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'sweep launch ms', d['roofline']['avg_launch_ms'], 'generators', d['config']['generators_found'])"
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats -d /tmp/p2 -o t -- python3 bench.py --workload gf2 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python3 profiles/summarize_rocpd.py /tmp/p2/t_results.db | head -5
