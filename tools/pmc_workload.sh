# HBM traffic of the dominant kernel of one bench.py workload (run on the GPU box):
#   bash tools/pmc_workload.sh rNN mul_cleanup|rotation|gf2|adjacency
# Two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in their own runs, no trace flags next to --pmc) and a kernel-trace pass of
# `python3 bench.py --workload W --steps 2 --warmup 1 --no-extras --no-cpu`.  Writes into gpurun_out/<tag>_<W>/:
#   <tag>_<short>_traffic.json  (read by bench.py; tied to the sha256 of the kernel source), <tag>_<short>_pmc.txt, <tag>_<short>_kernel_trace.txt,
#   <tag>_<short>_n1.json (the bench line under the profiler)
export TMPDIR=/tmp
tag=${1:-r05}; wl=${2:-rotation}
case $wl in
  mul_cleanup) short=cfg3; like='%k_emit_fused%'; srcs="cleanup.hip";;
  rotation)    short=rotation; like='%k_rot_resident%'; srcs="rotate_resident.hip";;
  gf2)         short=gf2; like='%k_sweep_m4r<1>%'; srcs="gf2.hip";;
  adjacency)   short=adjacency; like='%k_commutes_m4r%'; srcs="commute_m4r.hip commute_m4r7.hip";;
  *) echo "unknown workload $wl"; exit 2;;
esac
out=gpurun_out/${tag}_$wl; rm -rf $out; mkdir -p $out
cmd="bench.py --workload $wl --steps 2 --warmup 1 --no-extras --no-cpu --no-api"
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $out/w -o w -- python3 $cmd > $out/w.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $out/r -o r -- python3 $cmd > $out/r.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $out/t -o t -- python3 $cmd > $out/${tag}_${short}_n1.json 2> $out/t.log
python3 profiles/summarize_rocpd.py $out/t/t_results.db | head -24 > $out/${tag}_${short}_kernel_trace.txt
python3 profiles/summarize_rocpd.py --pmc $out/w/w_results.db --pmc $out/r/r_results.db | grep -E "^#|counter|k_emit|k_rot_res|k_sweep|k_heads|k_rs_|k_commutes|k_select|k_mul_coeff|k_mark|k_fixup|k_find" > $out/${tag}_${short}_pmc.txt
python3 - "$out" "$tag" "$wl" "$short" "$like" $srcs <<'PY'
import sqlite3, json, sys, hashlib
out, tag, wl, short, like = sys.argv[1:6]
srcs = sys.argv[6:]
def per_launch(db, counter):
    cur = sqlite3.connect(db).cursor()
    n, s, d = cur.execute("select count(*), sum(value), avg(duration) from counters_collection where counter_name=? and kernel_name like ?", (counter, like)).fetchone()
    return n, (s or 0.0) / max(1, n), d
nw, w, dw = per_launch(f'{out}/w/w_results.db', 'WRITE_SIZE')
nr, r, dr = per_launch(f'{out}/r/r_results.db', 'FETCH_SIZE')
line = [l for l in open(f'{out}/{tag}_{short}_n1.json') if l.startswith('{')][-1]
cfg = json.loads(line)['config']
config = {'mul_cleanup': lambda: {'workload': 'mul_cleanup', 'n_qubits': cfg['n_qubits'], 'terms': cfg['terms']},
          'rotation': lambda: {'workload': 'rotation', 'n_qubits': cfg['n_qubits'], 'terms': cfg['terms']},
          'gf2': lambda: {'workload': 'gf2', 'rows': cfg['matrix'][0], 'cols': cfg['matrix'][1]},
          'adjacency': lambda: {'workload': 'adjacency', 'n_qubits': cfg['n_qubits'], 'terms': cfg['terms']}}[wl]()
h = hashlib.sha256()
for s in srcs:
    h.update(open(f'symmer_amd/csrc/{s}', 'rb').read())
doc = {
    'source': f'rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) on `python3 bench.py --workload {wl} --steps 2 --warmup 1 --no-extras --no-cpu --no-api`, MI355X; summaries in profiles/{tag}_{short}_pmc.txt (tools/pmc_workload.sh)',
    'config': config, 'kernel': like.strip('%'), 'launches_profiled': nw, 'avg_launch_us_under_pmc': (dw or 0) / 1e3,
    'kernel_source_sha256': h.hexdigest(),
    'write_bytes_per_launch': int(w * 1024), 'fetch_bytes_per_launch_raw': int(r * 1024), 'fetch_bytes_per_launch_corrected_x2': int(2 * r * 1024),
    'note': 'counter unit is KiB. FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are tallied at half; Infinity-Cache hits are counted, so a working set that lives in the 256 MB Infinity Cache shows as fetch traffic although it does not reach the HBM).',
}
json.dump(doc, open(f'{out}/{tag}_{short}_traffic.json', 'w'), indent=2)
print(json.dumps(doc, indent=1))
PY
