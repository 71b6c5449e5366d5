#!/bin/bash
# tests, then two bench lines (20 and 300 timed steps) into gpurun_out/$1
d=gpurun_out/${1:-r4x}
mkdir -p $d
python -m pytest tests -m gpu -x -q > $d/tests.log 2>&1
echo "tests rc $?" > $d/rc.txt
tail -3 $d/tests.log
python3 bench.py --steps 20 --warmup 5 > $d/bench20.json 2> $d/bench20.err
python3 bench.py --steps 300 --warmup 20 > $d/bench300.json 2> $d/bench300.err
python3 - <<PY
import json
for f in ("$d/bench20.json", "$d/bench300.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
