#!/bin/bash
# interleaved bench repeats at several batch sizes for a list of library suffixes: r4_ab7.sh <out> "<sizes>" <sfx>...
out=$1; sizes=$2; shift 2
mkdir -p $out
P=trex-gym_amd/trex_gym
for rep in 1 2; do
  for n in $sizes; do
    for sfx in "$@"; do
      lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
      TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --envs-per-gpu $n --steps 200 --warmup 20 > $out/b$n$tag.$rep.json 2>> $out/err.log
    done
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
