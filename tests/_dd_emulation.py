"""Oracle-driven emulation of the N-rank BiCGStab of the domain-decomposed path (test infrastructure: uses oracle/).  Shared by
tests/test_gpu_distributed.py (in-process ranks) and tests/xrank_worker.py (one process per rank, consumer-side all-reduce)."""
import numpy as np


def oracle_rank(o, ck, sub, nz, b, perm, bp):
    """One rank of the emulation from what the device rank reports: its subdomain (dd.local_subdomain), assembled values and
    right-hand side (ghost rows already -I / owner-consistent), device ordering (perm, block pointers)."""
    nl, no = sub["n_local"], sub["n_owned"]
    osys = o.TPFASystem(sub["N"], nl)   # the host pattern (bit-exact with the device's, a-3)
    rp_p, ci_p, nz_p, part_p, _ = ck.device_order_problem(nl, 1, osys.rowptr, osys.colidx, nz, perm, bp)
    F = o.ILU0(nl, 1, rp_p, ci_p, nz_p, partition=part_p)
    return dict(nl=nl, no=no, rp=osys.rowptr, ci=osys.colidx, nz=nz, b=b, F=F, pe=perm - 1,
                send=sub["send"], recv=sub["recv"], neighbors=[int(q) for q in sub["neighbors"]])


def emulated_bicgstab(o, ranks, side, rtol, itmax=200):
    """Oracle-driven N-rank BiCGStab, sequential over the ranks (the twin of tests/test_dd_gloo.py's worker): per rank the local
    matrix on owned + ghost cells with ghost rows -I (linalg.jl:18-35), consistent!(x) before every mul! (linalg.jl:37-55), dots
    over owned entries summed over the ranks (krylov.jl:51-105), preconditioner = ONE block-Jacobi ILU(0) per rank in the device's
    elimination order and block partition with the ghost part of its input zeroed (parray_preconditioner_apply!, linalg.jl:78-88).
    ranks: list of dict(nl, no, rp, ci, nz, b, F (oracle ILU0 in device order), pe (device order -> local), cells, send, recv,
    neighbors).  Returns (residual history, iterations)."""
    R = len(ranks)

    def exchange(vs):  # consistent!: every ghost takes its owner's value
        for k, rk in enumerate(ranks):
            for q, rcv in zip(rk["neighbors"], rk["recv"]):
                j = list(ranks[q]["neighbors"]).index(k)
                vs[k][np.asarray(rcv) - 1] = vs[q][np.asarray(ranks[q]["send"][j]) - 1]

    def gdot(a, b):
        return float(sum(a[k][: rk["no"]] @ b[k][: rk["no"]] for k, rk in enumerate(ranks)))

    def prec(vs):
        out = []
        for rk, v in zip(ranks, vs):
            w = v.copy()
            w[rk["no"]:] = 0.0
            y = np.zeros_like(w)
            y[rk["pe"]] = rk["F"].apply(w[rk["pe"]])
            out.append(y)
        return out

    def mul(vs):
        vs = [v.copy() for v in vs]
        exchange(vs)
        return [o.spmv(rk["nl"], 1, rk["rp"], rk["ci"], rk["nz"], v) for rk, v in zip(ranks, vs)], vs

    def axpy(a, xs, ys):
        return [y + a * x for x, y in zip(xs, ys)]

    b = [rk["b"].copy() for rk in ranks]
    exchange(b)
    left = side == "left"
    r = prec(b) if left else b
    p, c = [v.copy() for v in r], [v.copy() for v in r]
    x = [np.zeros(rk["nl"]) for rk in ranks]
    rho = gdot(c, r)
    hist = [np.sqrt(gdot(r, r))]
    it = 0
    while hist[-1] > rtol * hist[0] and it < itmax:
        it += 1
        if left:
            y = p
            q = prec(mul(p)[0])
        else:
            y = prec(p)
            q, y = mul(y)          # (the exchanged copy: ghosts of y carry the owners' values, as on the device)
        alpha = rho / gdot(c, q)
        s = axpy(-alpha, q, r)
        if left:
            z = s
            t = prec(mul(s)[0])
        else:
            z = prec(s)
            t, z = mul(z)
        omega = gdot(t, s) / gdot(t, t)
        x = axpy(omega, z, axpy(alpha, y, x))
        r = axpy(-omega, t, s)
        rho_n = gdot(c, r)
        hist.append(np.sqrt(gdot(r, r)))
        beta = (rho_n / rho) * (alpha / omega)
        p = [rv + beta * (pv - omega * qv) for rv, pv, qv in zip(r, p, q)]
        rho = rho_n
    return np.array(hist), it, x
