# rocprofv3 summaries of the secondary kernels for profiles/ (run on the GPU box): bash tools/prof_round.sh r02
# kernel traces (--kernel-trace --stats) and, in separate runs, the LDS counters of the commutation kernel.
export TMPDIR=/tmp
tag=${1:-r02}
out=gpurun_out/${tag}x; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out/adj -o t -- python3 bench.py --workload adjacency --steps 2 --warmup 1 > $out/${tag}_adjacency_n1.json 2> $out/adj.log
python3 profiles/summarize_rocpd.py $out/adj/t_results.db | head -14 > $out/${tag}_adjacency_kernel_trace.txt
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $out/adjpmc -o p -- python3 bench.py --workload adjacency --steps 1 --warmup 0 > /dev/null 2> $out/adjpmc.log
python3 profiles/summarize_rocpd.py --pmc $out/adjpmc/p_results.db | grep -E "counter|k_commutes_m4r" > $out/${tag}_adjacency_lds_pmc.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $out/cfg3 -o t -- python3 tools/bench_kernels.py cfg3 > $out/cfg3.out 2> $out/cfg3.log
{ grep cfg3 $out/cfg3.out; python3 profiles/summarize_rocpd.py $out/cfg3/t_results.db | head -16; } > $out/${tag}_cfg3_kernel_trace.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $out/rot -o t -- python3 tools/bench_rot.py > $out/rot.out 2> $out/rot.log
{ grep rotation $out/rot.out; python3 profiles/summarize_rocpd.py $out/rot/t_results.db | head -14; } > $out/${tag}_rotation_kernel_trace.txt
GF2_ONLY_CFG4=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/gf2 -o t -- python3 tools/bench_gf2.py > $out/gf2.out 2> $out/gf2.log
{ grep "n=" $out/gf2.out; python3 profiles/summarize_rocpd.py $out/gf2/t_results.db | head -10; } > $out/${tag}_gf2_kernel_trace.txt
ls -la $out/*.txt $out/*.json
