export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "cleanup or lazy_gate or cfg3 or flagged or sort or mul or chain or partition or indexed" 2>&1 | tail -3
bash tools/prof_cfg3_quick.sh 26 | cut -c1-150
