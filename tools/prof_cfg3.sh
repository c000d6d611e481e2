# per-kernel breakdown of the cfg3 fused product+cleanup (run on the GPU box): bash tools/prof_cfg3.sh
export TMPDIR=/tmp
rm -rf gpurun_out/cfg3prof; mkdir -p gpurun_out/cfg3prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/cfg3prof -o t -- python3 tests/_bench_kernels.py ${1:-cfg3} > gpurun_out/cfg3prof/out.txt 2>&1
python3 profiles/summarize_rocpd.py gpurun_out/cfg3prof/t_results.db | head -${2:-24}
grep -E "cfg3|rotation|product|commute" gpurun_out/cfg3prof/out.txt
