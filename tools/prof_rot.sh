# rocprofv3 kernel trace of the cfg2 rotation (run on the GPU box): bash tools/prof_rot.sh [tag]
export TMPDIR=/tmp
tag=${1:-rot}
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
python3 tools/bench_rot.py > $out/rot_plain.out 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/rot -o t -- python3 tools/bench_rot.py > $out/rot.out 2> $out/rot.log
{ echo "un-profiled: $(grep rotation $out/rot_plain.out)"; grep rotation $out/rot.out; python3 profiles/summarize_rocpd.py $out/rot/t_results.db | head -${2:-12}; } > $out/rot_kernel_trace.txt
cut -c1-140 $out/rot_kernel_trace.txt
