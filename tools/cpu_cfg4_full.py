# the CPU baseline of cfg4 at FULL size (the row-XOR rate of the reference's Python loop depends on row length and row
# count): _rref_binary of the 4000 x 54000 GF(2) matrix, NumPy restatement of reference operators/utils.py:292-315, one core.
# ~10 minutes.  Writes profiles/r03_cpu_cfg4_full.json (bench.py --workload gf2 reports it as `full_size_cached`).
import sys, os, time, json, platform, datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_np as onp
rng = np.random.default_rng(1238)
symp = rng.random((50000, 4000)) < 0.3
symp[:, :32] = False                                   # 32 planted symmetries (no Clifford scrambling: the loop's cost does not depend on it)
n = 2000
M = np.vstack([np.hstack([symp[:, n:], symp[:, :n]]), np.eye(2 * n, dtype=bool)]).T.copy()      # [2n, M + 2n]: the transposed matrix _cref_binary reduces
t0 = time.perf_counter(); red, nx = onp.rref_noswap(M, count_xors=True); t = time.perf_counter() - t0
model = 'unknown'
try:
    for line in open('/proc/cpuinfo'):
        if line.startswith('model name'):
            model = line.split(':', 1)[1].strip(); break
except OSError:
    pass
doc = {'what': '_rref_binary (no row swaps) of the cfg4 matrix, NumPy restatement of reference utils.py:292-315, single thread', 'rows': int(M.shape[0]), 'cols': int(M.shape[1]),
       'row_xors': int(nx), 'seconds': t, 'row_xors_per_s': nx / t, 'cpu_model': model, 'host_cores': os.cpu_count(), 'host': platform.node(),
       'date': datetime.datetime.utcnow().isoformat() + 'Z', 'kernel_dimension': int((~red[:, :50000].any(axis=1)).sum())}
out = os.path.join(ROOT, 'gpurun_out', 'r03_cpu_cfg4_full.json')
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(doc, open(out, 'w'), indent=1)
print(json.dumps(doc))
