#!/bin/bash
# digest (bitwise against the first suffix) + interleaved bench repeats for a list of library suffixes
out=$1; shift
mkdir -p $out
P=trex-gym_amd/trex_gym
for sfx in "$@"; do
  lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
  TREX_LIB=$lib timeout -k 10 300 python scripts/state_digest.py 300 4096 > $out/digest4096$tag.txt 2>&1
  TREX_LIB=$lib timeout -k 10 300 python scripts/state_digest.py 100 1000 > $out/digest1000$tag.txt 2>&1
done
md5sum $out/digest*.txt
for rep in 1 2 3; do
  for sfx in "$@"; do
    lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
    TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $out/b4096$tag.$rep.json 2>> $out/err.log
    TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --envs-per-gpu 256 > $out/b256$tag.$rep.json 2>> $out/err.log
  done
done
for sfx in "$@"; do
  lib=$P/libtrex_hip${sfx}.so; tag=${sfx:-_product}
  TREX_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --envs-per-gpu 32768 --steps 100 > $out/b32768$tag.1.json 2>> $out/err.log
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
