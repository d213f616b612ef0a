#!/usr/bin/env python3
"""Times the host set-up of the library on a box WITHOUT a GPU, through a planning context (jh_context_create_host): device
ordering (weighted bisection), CSR pattern, tiles, jagged-slice layout, ILU(0) symbolic phase and layouts -- everything
`bench.py` reports as setup_s except the uploads.  Phases are printed by the library itself (option setup_timing = 1).

    python tools/setup_probe.py --cells 10000000 [--law twophase] [--mesh lattice|cartesian|delaunay|polyhedral] [--threads 16]
"""
import argparse
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=10_000_000)
    ap.add_argument("--mesh", default="lattice")
    ap.add_argument("--law", default="poisson")
    ap.add_argument("--grading", type=float, default=3.0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--heap", action="store_true", help="library option setup_heap = 1 during the set-up (freed heap memory is kept and reused)")
    ap.add_argument("--checksum", action="store_true", help="option plan_checksum = 1: print the 64-bit checksum of every table a device would be handed (bit-identity of two builds)")
    ap.add_argument("--cache", default="", help="npz file the generated mesh is kept in between runs (N, T, nc, nf)")
    args = ap.parse_args()
    if args.threads:
        os.environ["JH_SETUP_THREADS"] = str(args.threads)
    import bench
    import jutul_amd as ja
    t0 = time.time()
    if args.cache and os.path.exists(args.cache):
        import numpy as np
        z = np.load(args.cache)
        mesh, desc = dict(N=z["N"], T=z["T"], nc=int(z["nc"]), nf=int(z["nf"])), f"cached {args.cache}"
    else:
        mesh, desc = bench.make_mesh(ja, args, args.cells)
        if args.cache:
            import numpy as np
            np.savez(args.cache, N=mesh["N"], T=mesh["T"], nc=mesh["nc"], nf=mesh["nf"])
    print(f"mesh: {desc}, nc={mesh['nc']} nf={mesh['nf']} in {time.time() - t0:.2f} s", flush=True)
    N = 2 if args.law == "twophase" else 1
    for rep in range(args.repeat):
        ctx = ja.HIPContext("host", setup_timing=0 if args.no_timing else 1, setup_heap=1 if args.heap else 0, plan_checksum=1 if args.checksum else 0)
        f0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
        t0 = time.time()
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, mesh["N"], mesh["nc"], block_n=N, reorder="blocks", face_weights=mesh["T"])
        t1 = time.time()
        A = ja.StaticSparsityMatrixCSR(disc)
        info = A.spmv_info()
        t2 = time.time()
        prec = ja.ILUZeroPreconditioner(partition="blocks")
        prec.symbolic(A)
        ctx.set_option("setup_heap", 0)
        t3 = time.time()
        print(f"[probe] page faults {(resource.getrusage(resource.RUSAGE_SELF).ru_minflt - f0) * 4 // 1024} MB (4 KB pages), setup_heap {int(args.heap)}")
        print(f"[probe] discretisation {t1 - t0:.3f} s   jagged layout {t2 - t1:.3f} s   ilu symbolic {t3 - t2:.3f} s   total {t3 - t0:.3f} s"
              f"   (spmv_info {info}, ilu {prec.info()})", flush=True)
        if args.checksum:
            import zlib
            import numpy as np
            perm, bp = disc.ordering()
            print(f"[probe] plan_checksum {ctx.plan_checksum()}  ordering crc {zlib.crc32(np.ascontiguousarray(perm).tobytes())}  blocks {len(bp) - 1}", flush=True)
        del prec, A, disc, ctx


if __name__ == "__main__":
    main()
