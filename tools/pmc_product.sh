# HBM traffic of the dominant kernel of bench.py's workload (run on the GPU box): bash tools/pmc_product.sh rNN
# Two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit in one; counters in their own runs, no trace flags next to
# --pmc), then a kernel-trace pass of the same command.  Writes profiles/<tag>_traffic.json (read by bench.py, tied to the
# sha256 of product.hip), profiles/<tag>_product_pmc.txt and profiles/<tag>_bench_n1_kernel_trace.txt into gpurun_out/<tag>/.
export TMPDIR=/tmp
tag=${1:-r05}
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
cmd="bench.py --steps 2 --warmup 1 --no-extras --no-cpu"
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $out/w -o w -- python3 $cmd > $out/w.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $out/r -o r -- python3 $cmd > $out/r.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $out/t -o t -- python3 $cmd > $out/${tag}_bench_n1_under_rocprof.json 2> $out/t.log
python3 profiles/summarize_rocpd.py $out/t/t_results.db > $out/${tag}_bench_n1_kernel_trace.txt
python3 profiles/summarize_rocpd.py --pmc $out/w/w_results.db --pmc $out/r/r_results.db > $out/${tag}_product_pmc.txt
python3 - "$out" "$tag" <<'PY'
import sqlite3, json, sys, hashlib
out, tag = sys.argv[1], sys.argv[2]
def per_launch(db, counter, like):
    cur = sqlite3.connect(db).cursor()
    n, s = cur.execute("select count(*), sum(value) from counters_collection where counter_name=? and kernel_name like ?", (counter, like)).fetchone()
    return n, (s or 0.0) / max(1, n)
nw, w = per_launch(f'{out}/w/w_results.db', 'WRITE_SIZE', '%k_mul_rows_e%')
nr, r = per_launch(f'{out}/r/r_results.db', 'FETCH_SIZE', '%k_mul_rows_e%')
_, cw = per_launch(f'{out}/w/w_results.db', 'WRITE_SIZE', '%k_probe_copy%')
_, cr = per_launch(f'{out}/r/r_results.db', 'FETCH_SIZE', '%k_probe_copy%')
line = [l for l in open(f'{out}/{tag}_bench_n1_under_rocprof.json') if l.startswith('{')][-1]
cfg = json.loads(line)['config']
doc = {
    'source': f'rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) on `python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu`, MI355X; summaries in profiles/{tag}_product_pmc.txt (tools/pmc_product.sh)',
    'config': {'n_qubits': cfg['n_qubits'], 'left_terms_per_gpu': cfg['left_terms_per_gpu'], 'right_terms': cfg['right_terms'], 'slab_rows': cfg['slab_rows']},
    'kernel': 'k_mul_rows_e', 'launches_profiled': nw,
    'kernel_source_sha256': hashlib.sha256(open('symmer_amd/csrc/product.hip', 'rb').read()).hexdigest(),
    'write_bytes_per_launch': int(w * 1024), 'fetch_bytes_per_launch_raw': int(r * 1024), 'fetch_bytes_per_launch_corrected_x2': int(2 * r * 1024),
    'calibration_copy_4GiB': {'WRITE_SIZE_bytes': int(cw * 1024), 'FETCH_SIZE_bytes_raw': int(cr * 1024)},
    'note': 'counter unit is KiB. FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are tallied at half; Infinity-Cache hits are counted); the calibration entry is the library\'s own 4 GiB copy probe in the same runs.',
}
json.dump(doc, open(f'{out}/{tag}_traffic.json', 'w'), indent=2)
print(json.dumps(doc, indent=1))
PY
