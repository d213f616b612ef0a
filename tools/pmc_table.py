"""Per-kernel per-launch averages of rocprofv3 --pmc counter_collection CSVs (any number of passes).
usage: pmc_table.py out.json pass1.csv pass2.csv ...
Adds derived rows: hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE [KB -> B] (gfx950 FETCH correction, MI355X_MICROARCH.md HBM),
L2 hit rate, waves in flight, wait fractions.  Launches that did nothing (the speculative Krylov iteration of a converged
solve returns at once) are dropped: a launch counts only if it ran >= 20% of the kernel's median SQ_WAVE_CYCLES / bytes."""
import collections
import csv
import hashlib
import json
import os
import re
import statistics
import sys

KNOWN = ["ilu_apply_chunked_kernel", "ilu_apply_jds_kernel", "ilu_apply_blocks_pf_kernel", "ilu_factor_lds_kernel", "ilu_factor_wave_kernel", "ilu_factor_diag_kernel", "ilu_factor_prog_kernel", "ilu_factor_rows_kernel",
         "spmv_pipe_kernel", "spmv_jds16_kernel", "spmv_jds_kernel", "spmv_tile_kernel", "jagged_copy_kernel", "assemble_pipe_kernel", "assemble_tile_kernel",
         "bicg_xr_dots_kernel", "bicg_reduce_publish_kernel", "bicg_init_kernel", "dot2_partial_kernel", "final_reduce_kernel"]


def short(name):
    for k in KNOWN:
        if k in name:
            m = re.search(re.escape(k) + r"<([^>]*)>", name)
            return k + ("<" + m.group(1) + ">" if m else "")
    return None


def derive(row):
    f, w = row.get("FETCH_SIZE"), row.get("WRITE_SIZE")
    if f is not None and w is not None:
        row["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    if "TCC_HIT_sum" in row and row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0) > 0:
        row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in row and row["SQ_WAVE_CYCLES"] > 0:
        wc = row["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in row:
                row["frac_" + c] = row[c] / wc
        if row.get("SQ_BUSY_CYCLES"):
            row["waves_per_busy_cycle"] = wc / row["SQ_BUSY_CYCLES"]
    if "TCP_TCC_READ_REQ_LATENCY_sum" in row and row.get("TCP_TCC_READ_REQ_sum"):
        row["tcp_tcc_read_latency_cycles"] = row["TCP_TCC_READ_REQ_LATENCY_sum"] / row["TCP_TCC_READ_REQ_sum"]
    return row


def finish(out, out_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "jutul.jl_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp", ".cpp")):
            h.update(open(os.path.join(d, fn), "rb").read())
    try:  # the machine code of the library the passes ran on: host-side changes leave it (and the validity of this table) alone
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from fatbin_hash import device_code_hash
        dev_hash = device_code_hash(os.path.join(os.path.dirname(d), "libjutul_hip.so"))[0]
    except Exception:  # noqa: BLE001
        dev_hash = None
    out["_meta"] = {"kernel_source_hash": h.hexdigest()[:16], "device_code_hash": dev_hash, "units": "FETCH_SIZE/WRITE_SIZE in KB per launch; SQ_* cycles are quad-cycles summed over waves"}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for k, row in out.items():
        if k == "_meta":
            continue
        print("==", k)
        for c in sorted(row):
            if not c.startswith("launches_"):
                print(f"   {c:36s} {row[c]:16.4f}   (n={row.get('launches_' + c, '')})")


def main():
    if sys.argv[1] == "--merge":  # per-pass summaries -> one table
        out = collections.defaultdict(dict)
        for p in sys.argv[3:]:
            for k, row in json.load(open(p)).items():
                if k != "_meta":
                    out[k].update(row)
        finish({k: derive(r) for k, r in sorted(out.items())}, sys.argv[2])
        return
    out_path, paths = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> per-launch values
    for p in paths:
        per = collections.defaultdict(lambda: collections.defaultdict(float))  # (kernel, dispatch) -> counter -> value (summed over dims)
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                vals[k][c].append(v)
    out = {}
    for k, cs in sorted(vals.items()):
        row = {}
        for c, v in cs.items():
            med = statistics.median(v)
            keep = [x for x in v if x >= 0.2 * med] if med > 0 else v
            row[c] = sum(keep) / len(keep)
            row["launches_" + c] = len(keep)
        out[k] = derive(row)
    finish(out, out_path)


main()
