cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in main olddiv; do if [ $v != main ]; then export ARTIST_HIP_LIB=$PWD/tools/bin/libvar_$v.so; else unset ARTIST_HIP_LIB; fi; timeout -k 10 300 python tools/blocking_bench.py 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k:(round(v['fwd_ms'],2), round(v['fwd_bwd_ms'],2)) for k,v in d.items() if isinstance(v,dict)})"; done; done
unset ARTIST_HIP_LIB
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
