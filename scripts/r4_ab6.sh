#!/bin/bash
# interleaved 4096-env bench repeats (300 timed steps) for a list of library suffixes
out=$1; shift
mkdir -p $out
P=trex-gym_amd/trex_gym
for rep in 1 2 3; do
  for sfx in "$@"; do
    lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
    TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $out/b4096$tag.$rep.json 2>> $out/err.log
  done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$out/b*.json")):
    try:
        d = json.load(open(f)); name = f.split("/")[-1].rsplit(".", 2)[0]
        acc[name].append((d["value"] / 1e6, d["roofline"]["kernel_ms"]))
    except Exception as e:
        print(f, "FAILED", e)
for k, v in sorted(acc.items()):
    print("%-24s" % k, "  ".join("%.3f M / %.4f ms" % x for x in v))
PY
