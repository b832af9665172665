"""Write tests/golden/oracle_solver_stack2x10.npz: a regression fixture produced by the ORACLE (not by
the reference, whose solver translation unit cannot be built here).  It pins the oracle's own behaviour
across edits; parity with the reference for this path remains "unpinned".

    python tests/golden/make_oracle_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob  # noqa: E402
from helpers import presolve_state  # noqa: E402
from phyx_amd import scenes  # noqa: E402

b, cp, j = presolve_state(scenes.stack(2, 10), 2)
out = {"bodies": b.view(np.uint8), "cps": cp.view(np.uint8), "joints": j.view(np.uint8)}
for name, mode in (("scalar", 0), ("sse2", 1), ("avx2", 2)):
    bb, jj = b.copy(), j.copy()
    ob.solver_solve(bb, cp, jj, mode, ob.ISLAND_SINGLE, 15, 15)
    out["bodies_out_" + name] = bb.view(np.uint8)
    out["joints_out_" + name] = jj.view(np.uint8)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_solver_stack2x10.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
