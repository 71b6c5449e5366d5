#!/bin/bash
# A/B of step-kernel builds on one MI355X: bitwise digests against the round-3 build, then bench lines.
# usage: scripts/r4_ab.sh <outdir> <suffix> [<suffix> ...]   (suffix "" = the product library)
set -e
out=$1; shift
mkdir -p $out
P=trex-gym_amd/trex_gym
for sfx in "$@"; do
  lib=$P/libtrex_hip${sfx}.so
  tag=${sfx:-_product}
  TREX_LIB=$lib python scripts/state_digest.py 300 4096 > $out/digest4096$tag.txt 2>&1
  TREX_LIB=$lib python scripts/state_digest.py 100 8192 > $out/digest8192$tag.txt 2>&1
  TREX_LIB=$lib python bench.py --no-cpu-baseline > $out/bench$tag.json 2> $out/bench$tag.err
  TREX_LIB=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/bench20$tag.json 2>> $out/bench$tag.err
  TREX_LIB=$lib python bench.py --no-cpu-baseline --envs-per-gpu 256 > $out/bench256$tag.json 2>> $out/bench$tag.err
  TREX_LIB=$lib python bench.py --no-cpu-baseline --envs-per-gpu 8192 --steps 200 > $out/bench8192$tag.json 2>> $out/bench$tag.err
  TREX_LIB=$lib python bench.py --no-cpu-baseline --envs-per-gpu 32768 --steps 100 > $out/bench32768$tag.json 2>> $out/bench$tag.err
  python - <<PY
import json
for n in ("", "20", "256", "8192", "32768"):
    try:
        d = json.load(open("$out/bench%s$tag.json" % n))
        print("$tag", n or "4096x300", "%.3f M  kernel_ms %.4f" % (d["value"] / 1e6, d["roofline"]["kernel_ms"]))
    except Exception as e:
        print("$tag", n, "FAILED", e)
PY
done
md5sum $out/digest4096*.txt $out/digest8192*.txt
