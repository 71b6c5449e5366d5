#!/bin/bash
# interleaved repeats: does a build differ from another by more than the run-to-run spread?
out=$1; shift
mkdir -p $out
P=trex-gym_amd/trex_gym
for rep in 1 2 3; do
  for sfx in "$@"; do
    lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
    TREX_LIB=$lib python bench.py --no-cpu-baseline > $out/b4096$tag.$rep.json 2>> $out/err.log
    TREX_LIB=$lib python bench.py --no-cpu-baseline --envs-per-gpu 256 > $out/b256$tag.$rep.json 2>> $out/err.log
  done
done
for sfx in "$@"; do
  lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
  for n in 8192 16384 32768; do
    TREX_LIB=$lib python bench.py --no-cpu-baseline --envs-per-gpu $n --steps 100 > $out/b$n$tag.json 2>> $out/err.log
    TREX_BENCH_BALANCE=2 TREX_LIB=$lib python bench.py --no-cpu-baseline --envs-per-gpu $n --steps 100 > $out/b$n${tag}_mode2.json 2>> $out/err.log
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/b*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], "%.3f M  kernel_ms %.4f" % (d["value"] / 1e6, d["roofline"]["kernel_ms"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
