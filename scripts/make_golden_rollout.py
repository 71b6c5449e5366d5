#!/usr/bin/env python3
"""Generate tests/golden/oracle_rollout.npz from the f64 CPU oracle (oracle/trex_oracle.c).

pybullet is absent, so these are NOT pybullet outputs (parity unpinned): they freeze the
oracle's own trajectories so that (a) an accidental change of the oracle is caught by the CPU suite
and (b) the GPU suite has fixed vectors to compare with. Rollouts (BASELINE config 1 shapes):
  zero   - reset, then 40 steps of the zero action (reference plumbing case)
  crouch - reset, then 60 steps holding the start pose (free fall ~28 steps, then landing)
  random - reset, then 40 steps of uniform random actions (rng seed 0)
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import oracle as O, trex_model as tm  # noqa: E402


def main():
    m = tm.compile_model(O.default_asset_urdf())
    orc = O.Oracle(m)
    lo, hi = m["q_lower"][m["obs_order"]], m["q_upper"][m["obs_order"]]
    q0 = m["q_start"][m["obs_order"]]
    rng = np.random.default_rng(0)
    plans = {
        "zero": np.zeros((40, 25)),
        "crouch": np.tile(q0, (60, 1)),
        "random": rng.uniform(lo, hi, (40, 25)),
    }
    out = {}
    for name, acts in plans.items():
        s = orc.new_state()
        obs0 = orc.reset(s)
        obs, rew, pen, st = [obs0], [], [], [orc.get_state(s)]
        for a in acts:
            o, r, p = orc.step(s, a)
            obs.append(o); rew.append(r); pen.append(p); st.append(orc.get_state(s))
        out[name + "_actions"] = acts
        out[name + "_obs"] = np.array(obs)
        out[name + "_reward"] = np.array(rew)
        out[name + "_penalties"] = np.array(pen)
        out[name + "_state"] = np.array(st)
    path = os.path.join(ROOT, "tests", "golden", "oracle_rollout.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
