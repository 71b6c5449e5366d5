#!/bin/bash
# interleaved repeats at 4096 envs (+ 2048, 256) for a list of library suffixes; digest of each against the first
out=$1; shift
mkdir -p $out
P=trex-gym_amd/trex_gym
for sfx in "$@"; do
  lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
  TREX_LIB=$lib timeout -k 10 300 python scripts/state_digest.py 200 4096 2>&1 | grep -v amdgpu > $out/digest$tag.txt
done
md5sum $out/digest*.txt
for rep in 1 2 3; do
  for sfx in "$@"; do
    lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
    TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $out/b4096$tag.$rep.json 2>> $out/err.log
  done
done
for sfx in "$@"; do
  lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
  for n in 256 2048 32768; do
    TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --envs-per-gpu $n --steps 100 > $out/b$n$tag.1.json 2>> $out/err.log
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
