"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel per-launch averages.
usage: pmc_summary.py out.json FETCH=dir1/x_counter_collection.csv WRITE=dir2/x_counter_collection.csv"""
import csv, collections, json, re, sys

KNOWN = ["ilu_apply_chunked_kernel", "ilu_apply_blocks_pf_kernel", "ilu_apply_blocks_kernel", "ilu_factor_lds_kernel",
         "spmv_pipe_kernel", "spmv_tile_kernel", "assemble_pipe_kernel", "assemble_tile_kernel", "bicg_xr_dots_kernel", "bicg_reduce_publish_kernel", "bicg_p_kernel",
         "bicg_s_kernel", "dot2_partial_kernel", "final_reduce_kernel"]


def short(name):
    for k in KNOWN:
        if k in name:
            m = re.search(re.escape(k) + r"<([^>]*)>", name)
            return k + ("<" + m.group(1) + ">" if m else "")
    return None


out = collections.defaultdict(dict)
for arg in sys.argv[2:]:
    tag, path = arg.split("=", 1)
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        out[k][tag + "_KB_per_launch"] = agg[k] / cnt[k]
        out[k]["launches_" + tag] = cnt[k]
for k, v in out.items():
    f, w = v.get("FETCH_KB_per_launch"), v.get("WRITE_KB_per_launch")
    if f is not None and w is not None:
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streams -> doubled (MI355X_MICROARCH.md, HBM)
        v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    print(k, {a: round(b, 1) for a, b in v.items()})
