# round-3 GPU job B: register chain — tests, timings, kernel trace
export TMPDIR=/tmp
out=gpurun_out/r03b; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rotate_resident.py -x -q -m gpu -k "register_chain" > $out/pytest_chain.log 2>&1; echo "pytest rc=$?" >> $out/pytest_chain.log
tail -3 $out/pytest_chain.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or rotat" > $out/pytest_parity_rot.log 2>&1; echo "pytest rc=$?" >> $out/pytest_parity_rot.log
tail -2 $out/pytest_parity_rot.log
for nch in 8 4; do echo "NCH $nch"; SYMGPU_CHAIN_NCH=$nch timeout 300 python3 tools/bench_chain3.py > $out/chain3_$nch.out 2>&1; cat $out/chain3_$nch.out; done
SYMGPU_CHAIN_NCH=${BEST_NCH:-4} timeout 600 rocprofv3 --kernel-trace --stats -d $out/ch -o t -- python3 tools/bench_chain3.py > $out/ch.out 2> $out/ch.log
python3 profiles/summarize_rocpd.py $out/ch/t_results.db | grep -E "calls|cchain_reg|rs_coop|permute" | cut -c1-150
