cd $GRAFT_REPO_ROOT
export ADJ_TERMS=200000
for o in 1 0; do
echo "== ORDER=$o"
SYMGPU_M4R_DBG2=1 SYMGPU_M4R_ORDER=$o timeout 600 python tools/bench_adj_share.py "X=1" 2>&1 | grep -v "^$" | tail -32 | awk 'NR<=8 || /call/'
done
