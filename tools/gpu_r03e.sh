# round-3 GPU job E: GF(2) — tests, timing, kernel trace
export TMPDIR=/tmp
out=gpurun_out/r03e; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz2.py -x -q -m gpu -k "gf2 or rref or symgen or symmetry or cref or independent or generator" > $out/pytest_gf2.log 2>&1; echo "rc=$?" >> $out/pytest_gf2.log; tail -3 $out/pytest_gf2.log
timeout 300 python3 tools/bench_gf2.py > $out/gf2_plain.out 2>&1; cat $out/gf2_plain.out
GF2_ONLY_CFG4=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/gf2 -o t -- python3 tools/bench_gf2.py > $out/gf2.out 2> $out/gf2.log
{ grep "n=" $out/gf2.out; python3 profiles/summarize_rocpd.py $out/gf2/t_results.db | head -10; } > $out/gf2_kernel_trace.txt; cut -c1-150 $out/gf2_kernel_trace.txt
