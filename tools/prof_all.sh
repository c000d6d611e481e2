# Everything under profiles/ for one round (run on the GPU box): bash tools/prof_all.sh r04
# -> gpurun_out/<tag>_profiles/: bench lines (un-profiled) of all five workloads, rocprofv3 kernel traces, PMC passes, traffic JSONs
export TMPDIR=/tmp
tag=${1:-r06}
dst=gpurun_out/${tag}_profiles; rm -rf $dst; mkdir -p $dst
# 2. product: PMC + kernel trace (+ traffic JSON tied to product.hip)
bash tools/pmc_product.sh $tag > $dst/pmc_product.log 2>&1
cp gpurun_out/$tag/${tag}_traffic.json gpurun_out/$tag/${tag}_product_pmc.txt gpurun_out/$tag/${tag}_bench_n1_kernel_trace.txt gpurun_out/$tag/${tag}_bench_n1_under_rocprof.json $dst/ 2>/dev/null
# 3. the other workloads: PMC + kernel trace + traffic JSON
for wl in rotation mul_cleanup gf2 adjacency; do
  bash tools/pmc_workload.sh $tag $wl > $dst/pmc_$wl.log 2>&1
  cp gpurun_out/${tag}_$wl/${tag}_*_traffic.json gpurun_out/${tag}_$wl/${tag}_*_pmc.txt gpurun_out/${tag}_$wl/${tag}_*_kernel_trace.txt $dst/ 2>/dev/null
  cp gpurun_out/${tag}_$wl/${tag}_*_n1.json $dst/$(ls gpurun_out/${tag}_$wl/ | grep _n1.json | sed 's/_n1.json/_n1_under_rocprof.json/') 2>/dev/null
done
# 4. adjacency: kernel trace + LDS counters; run of Clifford rotations: kernel trace
timeout 600 rocprofv3 --kernel-trace --stats -d $dst/adj -o t -- python3 bench.py --workload adjacency --steps 2 --warmup 1 --no-api --no-cpu > $dst/adj_under_rocprof.json 2> $dst/adj.log
python3 profiles/summarize_rocpd.py $dst/adj/t_results.db | head -14 > $dst/${tag}_adjacency_kernel_trace.txt
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU -d $dst/adjpmc -o p -- python3 bench.py --workload adjacency --steps 1 --warmup 0 --no-api --no-cpu > /dev/null 2> $dst/adjpmc.log
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $dst/adjpmc2 -o p -- python3 bench.py --workload adjacency --steps 1 --warmup 0 --no-api --no-cpu > /dev/null 2> $dst/adjpmc2.log
python3 profiles/summarize_rocpd.py --pmc $dst/adjpmc/p_results.db --pmc $dst/adjpmc2/p_results.db | grep -E "counter|k_commutes_m4r" > $dst/${tag}_adjacency_lds_pmc.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $dst/chain -o t -- python3 tools/bench_clifford_run.py > $dst/chain.out 2> $dst/chain.log
{ grep chain $dst/chain.out; python3 profiles/summarize_rocpd.py $dst/chain/t_results.db | grep -E "calls|cchain_reg|rs_coop|permute|cchain_flags|cchain_move" ; } > $dst/${tag}_clifford_run_kernel_trace.txt
rm -rf $dst/adj $dst/adjpmc $dst/adjpmc2 $dst/chain
# the traffic JSONs of THIS source go where bench.py looks for them (on the box's copy of the tree), then:
cp $dst/${tag}_*traffic.json profiles/ 2>/dev/null
# 5. the bench lines as the driver would see them (un-profiled; their `traffic` comes from the JSONs just written)
timeout 900 python3 bench.py > $dst/${tag}_bench_n1.json 2> $dst/bench_n1.err
for wl in rotation mul_cleanup gf2 adjacency; do
  timeout 900 python3 bench.py --workload $wl > $dst/${tag}_${wl}_n1.json 2> $dst/${wl}_n1.err
done
ls -la $dst | head -50
