"""Dev tool: cost of one device-scalar read (jh_vec_dot on a tiny vector = 2 launches + the read) -- pinned-record spin vs
D2H copy + stream synchronise (JH_OPTIONS=read_sync=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
ctx = ja.HIPContext(0)
g = ja.tet_lattice_mesh(4, 4, 4)
disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"])
v = ja.DeviceVector(disc, np.ones(g["nc"]))
for _ in range(200): v.dot(v)
t0 = time.perf_counter()
n = 5000
for _ in range(n): v.dot(v)
print(f"read_sync={ctx.get_option('read_sync')}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per dot+read", flush=True)
