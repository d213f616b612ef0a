"""CPU-only: the C-ABI library loads and exports every symbol include/jutul_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "jutul_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint32_t\s+(jh_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    so = os.path.join(ROOT, "jutul.jl_amd", "libjutul_hip.so")
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    names = header_functions()
    assert len(names) >= 55
    for n in names:
        assert hasattr(lib, n), f"{n} declared in jutul_hip.h but not exported"


def test_python_binding_covers_header():
    import jutul_amd  # noqa: F401
    from jutul_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_error_path_without_gpu_is_loud():
    """No CPU fallback: creating a context without a GPU raises (and with a GPU it works)."""
    import torch
    import jutul_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(jutul_amd.JutulHIPError):
        jutul_amd.HIPContext(0)


def test_meshgen_tet_lattice_properties():
    from jutul_amd import tet_lattice_mesh
    g = tet_lattice_mesh(5, 4, 3)
    nc, nf, N = g["nc"], g["nf"], g["N"]
    assert nc == 6 * 60 and N.shape == (2, nf)
    assert nf == 6 * 60 + 2 * ((5 - 1) * 4 * 3 + 5 * (4 - 1) * 3 + 5 * 4 * (3 - 1))
    deg = np.bincount(N.reshape(-1), minlength=nc + 1)[1:]
    assert deg.max() == 4 and deg.min() >= 2
    # no duplicate cell pairs, no self loops
    key = np.minimum(N[0], N[1]) * (nc + 1) + np.maximum(N[0], N[1])
    assert np.unique(key).size == nf and np.all(N[0] != N[1])
    assert np.all(g["T"] > 0) and np.all(g["volumes"] > 0)
    assert np.isclose(g["volumes"].sum(), 5 * 4 * 3, rtol=0.2)
    # deterministic
    g2 = tet_lattice_mesh(5, 4, 3)
    assert np.array_equal(g2["N"], N) and np.array_equal(g2["T"], g["T"])


def test_meshgen_trans_matches_oracle(oracle):
    from jutul_amd import tet_lattice_mesh, cartesian_neighbors
    g = tet_lattice_mesh(3, 3, 2)
    geo = dict(nc=g["nc"], dim=3, cell_centroids=g["cell_centroids"], face_centroids=g["face_centroids"],
               normals=g["normals"], areas=g["areas"])
    h = oracle.half_face_map(g["N"], g["nc"])
    Thf = oracle.half_face_trans(geo, g["perm_k"][None, :], h)
    assert np.allclose(oracle.face_trans(Thf, h["faces"], g["nf"]), g["T"], rtol=1e-12)
    for dims in [(3, 3, 1), (4, 3, 5), (7,), (2, 6)]:
        assert np.array_equal(cartesian_neighbors(dims), oracle.cartesian_geometry(dims)["N"])


def test_header_is_valid_c99_and_a_c_program_drives_the_boundary(tmp_path):
    """include/jutul_hip.h compiles as pedantic C99 (and as C++); tests/c_abi_consumer.c -- a C program, not ctypes -- links against the
    library, gets the reference-exact tables of the Poisson 3x1 case through a planning context, is refused by a compute entry point and
    by a device that is not there."""
    import shutil
    import subprocess
    import __graft_entry__ as g
    g.build()
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this box")
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "jutul.jl_amd")
    probe = tmp_path / "hdr.c"
    probe.write_text('#include "jutul_hip.h"\nint main(void) { return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(probe)], check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-x", "c++", "-I", inc, "-fsyntax-only", str(probe)], check=True)
    exe = tmp_path / "consumer"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, os.path.join(ROOT, "tests", "c_abi_consumer.c"),
                    "-o", str(exe), "-L", libdir, "-ljutul_hip", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C_ABI_CONSUMER_OK" in r.stdout, r.stdout + r.stderr
