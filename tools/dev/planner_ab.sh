# A/B of planner library builds on bench.py's C4 / C5 legs: bash tools/dev/planner_ab.sh lib_a.so lib_b.so   (round 6: the conv epilogue's residual loads)
for rep in 1 2; do for lib in "$@"; do for w in c5 c4; do
  echo -n "$lib $w: "
  python - "$lib" $w <<'PY' 2>&1 | grep -v amdgpu | tail -1
import json, os, runpy, sys, io, contextlib
from muzero_amd import planner as pl
pl.LIB_PATH = os.path.abspath(sys.argv[1])
w = sys.argv[2]
sys.argv = ['bench.py', '--workload', w, '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-sustained']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path('bench.py', run_name='__main__')
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('%.1f k sims/s frac %.4f' % (d['value'] / 1e3, d['roofline']['frac']))
PY
done; done; done
