# per-kernel breakdown of any helper script (run on the GPU box): bash tools/prof_any.sh tools/bench_gf2.py [rows]
export TMPDIR=/tmp
rm -rf gpurun_out/anyprof; mkdir -p gpurun_out/anyprof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/anyprof -o t -- python3 "$1" > gpurun_out/anyprof/out.txt 2>&1
python3 profiles/summarize_rocpd.py gpurun_out/anyprof/t_results.db | head -${2:-16}
grep -v "^W2026\|^E2026" gpurun_out/anyprof/out.txt | tail -5
