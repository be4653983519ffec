R=$GRAFT_REPO_ROOT; cd $R
for v in m0 m_nomix m_noev m_nostore m_all; do echo $v; PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy_n.py 2>&1 | grep "N=" | grep -E "N=  8192|N= 32768"; done
