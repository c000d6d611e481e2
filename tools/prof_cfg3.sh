# rocprofv3 kernel trace of the cfg3 product + cleanup (run on the GPU box): bash tools/prof_cfg3.sh [tag]
export TMPDIR=/tmp
tag=${1:-cfg3}
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out/cfg3 -o t -- python3 tools/bench_kernels.py cfg3 > $out/cfg3.out 2> $out/cfg3.log
{ grep cfg3 $out/cfg3.out; python3 profiles/summarize_rocpd.py $out/cfg3/t_results.db | head -${2:-16}; } > $out/cfg3_kernel_trace.txt
cut -c1-150 $out/cfg3_kernel_trace.txt
