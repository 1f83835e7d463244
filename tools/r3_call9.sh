cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/ab_libs.py --rounds 5 --reps 3 main= prio1=tools/bin/libvar_prio1.so prio3=tools/bin/libvar_prio3.so 2>&1 | tail -8
