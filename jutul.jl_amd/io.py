"""Result / restart IO of the Simulator slice, host side.

Mirrors the reference's on-disk protocol (simulator/io.jl:52-91, utils.jl:701-727, simulator.jl:680-705,
ext/JutulPartitionedArraysExt/io.jl): one file per report step `jutul_<step>` holding {state, report, step}; a restart from
step k reads step k-1 as the new state0 and its last ministep's dt; distributed runs write per-rank folders `proc_<rank>`
(1-based) with a `partition` file and are consolidated into global states afterwards.  The container is numpy's `.npz`
instead of JLD2 (same keys); states are dicts name -> array with the cell index LAST (the reference's [N, nc] layout)."""
import glob
import json
import os
import re

import numpy as np


def initialize_io(path):
    if path is not None:
        os.makedirs(path, exist_ok=True)


def step_path(path, step):
    return os.path.join(path, f"jutul_{int(step)}.npz")


def write_result(path, state, report, step):
    """write_result_jld2 (simulator/io.jl:84-91)."""
    arrays = {"state/" + k: np.asarray(v) for k, v in state.items()}
    np.savez(step_path(path, step), step=np.int64(step), report=np.array(json.dumps(report)), **arrays)


def read_restart(path, step, read_state=True, read_report=True):
    """read_restart (utils.jl:701-727): (state, report) of one step; ({} , None) with a warning if the file is missing."""
    f = step_path(path, step)
    if not os.path.isfile(f):
        import warnings
        warnings.warn(f"Data for step {step} was requested, but no such file was found at {f}.")
        return {}, None
    with np.load(f, allow_pickle=False) as z:
        if int(z["step"]) != int(step):
            import warnings
            warnings.warn(f"File {f} contained step {int(z['step'])}, but was named as step {step}.")
        state = {k[len("state/"):]: z[k].copy() for k in z.files if k.startswith("state/")} if read_state else {}
        report = json.loads(str(z["report"])) if read_report else None
    return state, report


def valid_restart_indices(path):
    """Steps for which a result file exists."""
    out = []
    for f in glob.glob(os.path.join(path, "jutul_*.npz")):
        m = re.fullmatch(r"jutul_(\d+)\.npz", os.path.basename(f))
        if m:
            out.append(int(m.group(1)))
    return sorted(out)


def deserialize_restart(path, restart, nsteps):
    """deserialize_restart (simulator.jl:680-705).  restart: True (continue after the last stored step) or a 1-based step.
    Returns (state0 or None, dt or None, first_step)."""
    if path is None:
        raise ValueError("output_path must be specified if restarts are enabled")
    if restart is True:
        ix = valid_restart_indices(path)
        restart = min((max(ix) + 1) if ix else 1, nsteps + 1)
    if restart > nsteps + 1:
        raise ValueError(f"Restart was {restart} but schedule contains {nsteps} steps.")
    first_step = int(restart)
    state0, dt = None, None
    if first_step > 1:
        state0, report0 = read_restart(path, first_step - 1)
        if report0 is not None and report0.get("ministeps"):
            dt = report0["ministeps"][-1]["dt"]
    return state0, dt, first_step


# ---- distributed runs -----------------------------------------------------------------------------------------------------
def rank_folder(path, rank):
    """rank is 1-based like the reference's proc_<rank> folders (ext/.../io.jl:52-54)."""
    return os.path.join(path, f"proc_{int(rank)}")


def write_partition(path, rank, cells_global, n_owned, n_total):
    """Per-rank partition file: the global ids (1-based) of the rank's local cells, owned first."""
    d = rank_folder(path, rank)
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, "partition.npz"), cells=np.asarray(cells_global, dtype=np.int64), n_owned=np.int64(n_owned),
             n_total=np.int64(n_total))


def consolidate_distributed_results(states, reports, partitions):
    """Global state from the ranks' local ones: every rank contributes its OWNED cells (consolidate_cell_values!,
    ext/.../io.jl:56-110); the report is rank 1's (all ranks take the same steps)."""
    n_total = int(partitions[0]["n_total"])
    out = {}
    for k in states[0]:
        first = np.asarray(states[0][k])
        full = np.zeros(first.shape[:-1] + (n_total,), dtype=first.dtype)
        for st, p in zip(states, partitions):
            no = int(p["n_owned"])
            full[..., np.asarray(p["cells"][:no]) - 1] = np.asarray(st[k])[..., :no]
        out[k] = full
    return out, reports[0]


def consolidate_distributed_results_on_disk(path, nranks, steps, cleanup=True):
    """consolidate_distributed_results_on_disk! (ext/.../io.jl:1-37)."""
    parts = []
    for r in range(1, nranks + 1):
        with np.load(os.path.join(rank_folder(path, r), "partition.npz")) as z:
            parts.append({k: z[k].copy() for k in z.files})
    written = []
    for step in steps:
        sr = [read_restart(rank_folder(path, r), step) for r in range(1, nranks + 1)]
        state, report = consolidate_distributed_results([s for s, _ in sr], [rp for _, rp in sr], parts)
        write_result(path, state, report, step)
        written += [step_path(rank_folder(path, r), step) for r in range(1, nranks + 1)]
    if cleanup:
        for f in written:
            os.remove(f)
