"""Host-side sanitizer run of the CPU oracle (the reference has none): the C restatement is
rebuilt with -fsanitize=address,undefined and driven through reset + contact-rich steps in a child
process. GPU sanitizers are not available on this pool; the HIP path is covered by the parity tests."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O, trex_model as tm
import oracle.oracle as mod
mod.build = lambda force=False: [%(lib)r, %(lib)r]      # load the sanitized build
m = tm.compile_model(O.default_asset_urdf())
orc = O.Oracle(m)
s = orc.new_state()
orc.reset(s)
rng = np.random.default_rng(0)
lo, hi = m["q_lower"][m["obs_order"]], m["q_upper"][m["obs_order"]]
for t in range(60):
    orc.step(s, rng.uniform(lo, hi) if t %% 2 else m["q_start"][m["obs_order"]])
orc.minv(s); orc.energy(s); orc.contacts(s); orc.forward_dynamics(s)
print("SANITIZED_RUN_OK", len(orc.contacts(s)[0]))
"""


def test_oracle_under_asan_ubsan():
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/liboracle_asan.so"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=lib)], env=env, capture_output=True, text=True, timeout=300)
    assert "SANITIZED_RUN_OK" in r.stdout, r.stderr[-2000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    assert r.returncode == 0
