"""Dev tool: how local are the SpMV column indices in the device ordering?  Fraction of entries of a 256-row tile whose
column lies inside the window [r0 - W, r0 + 256 + W)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells
cells = int(os.environ.get("CELLS", "1250000"))
ctx = ja.HIPContext(0)
g = ja.tet_lattice_mesh(*dims_for_cells(cells)); nc = g["nc"]
for br in (256, 512):
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=br)
    perm = disc.ordering()[0] - 1         # perm[i] = host row at device position i
    iperm = np.empty(nc, dtype=np.int64); iperm[perm] = np.arange(nc)
    N = g["N"] - 1
    a = iperm[N[0]]; b = iperm[N[1]]
    rows = np.concatenate([a, b]); cols = np.concatenate([b, a])
    tile = rows // 256
    r0 = tile * 256
    for W in (0, 128, 256, 512, 1024, 4096):
        inside = (cols >= r0 - W) & (cols < r0 + 256 + W)
        print(f"block_rows {br} W {W:5d}: {inside.mean():.4f} of off-diagonal entries inside the window", flush=True)
    d = np.abs(cols - rows)
    print("  |col-row| percentiles 50/75/90/95/99:", np.percentile(d, [50, 75, 90, 95, 99]).astype(int))
