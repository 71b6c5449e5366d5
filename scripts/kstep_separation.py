"""How fast do the f32 kernel and the f64 oracle separate over K steps THROUGH CONTACT, against the oracle's own f32 build?
50 states sampled along a 300-step landing (the states of tests/test_gpu_parity.py::test_one_step_parity_in_contact_and_at_rest),
K steps of a fixed action sequence on the GPU, on the f64 oracle and on the f32 oracle:
    python scripts/kstep_separation.py [K ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from oracle import oracle as O, trex_model as tm  # noqa: E402
from trex_gym.vec_env import TrexVecEnv  # noqa: E402


def landing_states(o64, model, seed=5, every=6, steps=300):
    q0 = model["q_start"][model["obs_order"]]
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    rng = np.random.default_rng(seed)
    s = o64.new_state()
    o64.reset(s)
    states = []
    for t in range(steps):
        o64.step(s, np.clip(q0 + 0.15 * rng.normal(size=25), lo, hi))
        if t % every == 0:
            states.append(o64.get_state(s).astype(np.float32))
    return np.array(states)


def separations(K, states, model, o64, o32, dev="cuda:0"):
    """-> per state: (dq_gpu, dq_32, dqd_gpu, dqd_32) after K steps, qd on the scale max(1, |qd|_inf of the f64 run)"""
    q0 = model["q_start"][model["obs_order"]]
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    n = len(states)
    rng = np.random.default_rng(11)
    acts = np.clip(q0 + 0.15 * rng.normal(size=(K, n, 25)), lo, hi).astype(np.float32)
    v = TrexVecEnv(n, device=dev)
    v.reset()
    v.set_state(torch.tensor(states))
    for k in range(K):
        obs, _, _, _ = v.step(acts[k])
    out = []
    for e in range(n):
        res = []
        for orc in (o64, o32):
            s = orc.new_state()
            orc.set_state(s, states[e].astype(np.float64))
            for k in range(K):
                o, _, _ = orc.step(s, acts[k, e].astype(np.float64))
            res.append(o)
        a, b = res
        scale = max(1.0, np.abs(a[25:50]).max())
        out.append((np.abs(obs[e, :25] - a[:25]).max(), np.abs(b[:25] - a[:25]).max(),
                    np.abs(obs[e, 25:50] - a[25:50]).max() / scale, np.abs(b[25:50] - a[25:50]).max() / scale))
    return np.array(out)


if __name__ == "__main__":
    model = tm.compile_model(O.default_asset_urdf())
    o64, o32 = O.Oracle(model, precision="f64"), O.Oracle(model, precision="f32")
    st = landing_states(o64, model)
    for K in [int(x) for x in sys.argv[1:]] or [1, 5, 10]:
        s = separations(K, st, model, o64, o32)
        print("K = %2d  (%d states)" % (K, len(s)))
        for name, g, r in (("|dq| [rad]", s[:, 0], s[:, 1]), ("|dqd| / max(1,|qd|)", s[:, 2], s[:, 3])):
            ratio = g / np.maximum(r, 1e-12)
            print("  %-22s kernel-vs-f64: median %.2e max %.2e | oracle f32-vs-f64: median %.2e max %.2e | ratio per state: median %.2f p90 %.2f max %.2f"
                  % (name, np.median(g), g.max(), np.median(r), r.max(), np.median(ratio), np.percentile(ratio, 90), ratio.max()))
        worst = np.argsort(-(s[:, 2] / np.maximum(s[:, 3], 1e-12)))[:5]
        print("  worst qd ratios: " + "  ".join("state %d: %.2e vs %.2e" % (i, s[i, 2], s[i, 3]) for i in worst))
