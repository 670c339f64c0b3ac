cd ${GRAFT_REPO_ROOT:-/root/repo}
OVQE_LIB=testing python tools/exp_streams.py 12 5 sector_debug=8 2>&1 | grep "streams of a sweep" | head -50
