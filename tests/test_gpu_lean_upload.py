"""Option ilu_lean_upload (csrc/jh_ilu.hip, written in a session without a GPU; default 0): with the chunk-jagged layout and a factorisation
that writes it directly, the row-major ILU(0) structure arrays are not uploaded.  OPT-IN until it has run once:
    JH_TEST_LEAN=1 python -m pytest tests/test_gpu_lean_upload.py -m gpu -q
(and the whole suite under JH_OPTIONS=ilu_lean_upload=1).  The kernels and every other table are the same, so factors, applies and
residual histories must be bit-identical to the default."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("JH_TEST_LEAN") != "1", reason="opt-in: JH_TEST_LEAN=1 (option not yet validated on a GPU)")]


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


def newton_step(ja, lean, g, kind, nblk):
    ctx = ja.HIPContext(0, ilu_lean_upload=lean)
    nc = g["nc"]
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=nblk, reorder="blocks", face_weights=g["T"])
    law = ja.ConservationLaw(disc, kind, rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
    rng = np.random.default_rng(4)
    T = g["T"] / g["T"].mean()
    X = rng.uniform(1.0, 2.0, nc) if nblk == 1 else np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
    law.set_face_trans(T)
    law.set_volumes(g["volumes"])
    law.set_state(X)
    law.set_state0(X)
    law.set_sources([1, nc], list(np.resize([1.0, -1.0], 2 * nblk)) if nblk == 1 else [0.5, 0.5, -0.5, -0.5])
    prec = ja.ILUZeroPreconditioner(partition="blocks")
    ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-8, max_iterations=200)
    sim = ja.Simulator(law, ks)
    rep = sim.perform_step(0.7, 1)
    info = prec.info()
    return dict(state=law.get_state(), its=int(rep.linear_iterations), factors=prec.factor_values(), kernel=info["factor_kernel"], jag=info["jagged"])


@pytest.mark.parametrize("family,kind,nblk", [("lattice", "poisson", 1), ("lattice", "twophase", 2), ("delaunay", "compressible", 1),
                                               ("polyhedral", "compressible", 1), ("cartesian", "twophase", 2)])
def test_lean_upload_changes_nothing(ja, family, kind, nblk):
    g = {"lattice": lambda: ja.tet_lattice_mesh(24, 22, 20, scramble=True), "delaunay": lambda: ja.delaunay_tet_mesh(12000, grading=2.0),
         "polyhedral": lambda: ja.polyhedral_dual_mesh(20000, grading=1.5), "cartesian": lambda: ja.cartesian_mesh(40, 38, 36)}[family]()
    a = newton_step(ja, 0, g, kind, nblk)
    b = newton_step(ja, 1, g, kind, nblk)
    assert a["jag"] and a["kernel"] == b["kernel"]
    assert a["its"] == b["its"]
    assert np.array_equal(a["factors"], b["factors"])
    assert np.array_equal(a["state"], b["state"])
