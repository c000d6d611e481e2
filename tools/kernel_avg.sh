# average duration of the kernels matching $1 over 20 steps of a bench workload (GPU box): bash tools/kernel_avg.sh find_suspects [workload]
export TMPDIR=/tmp
rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats -d /tmp/pk -o t -- python3 bench.py --workload ${2:-mul_cleanup} --steps 20 --warmup 2 --no-cpu --no-api --no-extras > /dev/null 2>&1
python3 profiles/summarize_rocpd.py /tmp/pk/t_results.db | grep -E "$1" | cut -c1-110
