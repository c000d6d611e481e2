# round-3 GPU job G: radix sort without flat LDS accesses — parity + cfg3 / chain timings
export TMPDIR=/tmp
out=gpurun_out/r03g; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz2.py tests/test_gpu_rotate_resident.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 600 python3 bench.py --workload mul_cleanup --steps 5 --warmup 2 --no-cpu > $out/cfg3.json 2>$out/cfg3.err; python3 -c "
import json; d=json.load(open('$out/cfg3.json')); print('cfg3 ms/step', d['ms_per_step'], 'emit', d['roofline']['achieved'])"
timeout 300 python3 tools/bench_chain3.py 2>&1 | head -3
timeout 600 rocprofv3 --kernel-trace --stats -d $out/t -o t -- python3 bench.py --workload mul_cleanup --steps 2 --warmup 1 --no-cpu > /dev/null 2> $out/t.log
python3 profiles/summarize_rocpd.py $out/t/t_results.db | head -12 | cut -c1-130
