python - <<'PY'
import re,subprocess,sys
src=open('tools/flux_bench.py').read().replace("B, Hh, W = 1000, 256, 256","B, Hh, W = 125, 256, 256")
exec(compile(src,'flux_bench_small','exec'))
PY
