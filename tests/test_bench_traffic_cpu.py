"""bench.py quotes roofline.traffic from the committed PMC passes only when they were taken on the current kernel sources; this
pins the lookup itself (kernel-name matching against the newest profiles/r*_traffic_10M_1gpu.json), which needs no GPU."""
import glob
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_pmc_passes_are_found_for_every_bench_kernel(monkeypatch):
    import bench
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_10M_1gpu.json")))[-1]
    meta = json.load(open(newest))["_meta"]
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: meta["kernel_source_hash"])
    args = types.SimpleNamespace(law="poisson", mesh="lattice")
    for kernel, lo, hi in (("ilu0_apply", 0.8e9, 1.3e9), ("spmv", 0.6e9, 1.2e9), ("assembly", 1.0e9, 2.0e9), ("ilu0_factor", 0.8e9, 2.5e9)):
        traffic, note = bench.measured_traffic(kernel, args, 10_025_988, 1)
        assert traffic is not None and lo < traffic < hi, (kernel, traffic, note)
        assert os.path.basename(newest) in note
    # another workload (none committed for it) or another tree: null with the reason
    assert bench.measured_traffic("spmv", args, 777_000, 1)[0] is None
    assert bench.traffic_tag(args, 10_025_988) == "10M" and bench.traffic_tag(types.SimpleNamespace(law="twophase", mesh="lattice"), 5) == "twophase_lattice_5"
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 16)
    traffic, note = bench.measured_traffic("spmv", args, 10_025_988, 1)
    assert traffic is None and "no committed PMC pass" in note
