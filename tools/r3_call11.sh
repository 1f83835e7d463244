cd $GRAFT_REPO_ROOT
ARTIST_HIP_PRINT_GEOMETRY=1 timeout -k 10 300 python tools/debug_flake.py 2>&1 | grep -v "art_trace_fwd" | sort | uniq -c | sort -rn | head -30
