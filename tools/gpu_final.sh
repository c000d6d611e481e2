# whole GPU suite, then every profile of the round (run on the GPU box): bash tools/gpu_final.sh r03
export TMPDIR=/tmp
tag=${1:-r03}
mkdir -p gpurun_out/${tag}_final
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_final/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_final/pytest_all.log; grep -E "passed|failed|rc=" gpurun_out/${tag}_final/pytest_all.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_final/smoke.log 2>&1; tail -2 gpurun_out/${tag}_final/smoke.log
bash tools/prof_all.sh $tag > gpurun_out/${tag}_final/prof_all.log 2>&1; tail -3 gpurun_out/${tag}_final/prof_all.log
