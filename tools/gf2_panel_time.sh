# duration of the panel as its own kernel (no look-ahead schedule), lean and generic loop (GPU box)
export TMPDIR=/tmp
for lean in 1 0; do
rm -rf /tmp/p3; SYMGPU_GF2_LOOKAHEAD=0 SYMGPU_GF2_LEAN_PANEL=$lean rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t -- python3 bench.py --workload gf2 --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
echo lean=$lean; python3 profiles/summarize_rocpd.py /tmp/p3/t_results.db | grep -E "wpanel|k_select|k_sweep|k_lead"
done
