# round-3 GPU job D: full-size parity of what the bench runs + the whole GPU suite
export TMPDIR=/tmp
out=gpurun_out/r03d; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "every_slab or whole_adjacency" > $out/pytest_full.log 2>&1; echo "rc=$?" >> $out/pytest_full.log; tail -5 $out/pytest_full.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_all.log 2>&1; echo "rc=$?" >> $out/pytest_all.log; tail -5 $out/pytest_all.log
