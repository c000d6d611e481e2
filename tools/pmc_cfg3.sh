# PMC counters of the cfg3 product + cleanup kernels (run on the GPU box): bash tools/pmc_cfg3.sh "COUNTER ..." [kernel filter]
export TMPDIR=/tmp
out=gpurun_out/pmc_cfg3; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc $1 -d $out/p -o p -- python3 tools/bench_kernels.py cfg3 > $out/p.out 2> $out/p.log
python3 profiles/summarize_rocpd.py --pmc $out/p/p_results.db | grep -E "counter|${2:-k_}" | cut -c1-160
