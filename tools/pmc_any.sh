# hardware counters of any helper script (run on the GPU box): bash tools/pmc_any.sh "CNT1 CNT2 ..." script.py [args...]
# (counters in their own run, no trace/stats flags next to --pmc — see the gpurun rules)
export TMPDIR=/tmp
cnt="$1"; shift
rm -rf gpurun_out/pmcany; mkdir -p gpurun_out/pmcany
timeout 600 rocprofv3 --pmc $cnt -d gpurun_out/pmcany -o p -- python3 "$@" > gpurun_out/pmcany/out.txt 2>&1
python3 profiles/summarize_rocpd.py --pmc gpurun_out/pmcany/p_results.db | grep -v "^$" | head -60
grep -v "^W2026\|^E2026" gpurun_out/pmcany/out.txt | tail -4
