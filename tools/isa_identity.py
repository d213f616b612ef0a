#!/usr/bin/env python3
"""Per-KERNEL identity of the device code of two builds of libjutul_hip.so.

tools/fatbin_hash.py answers "are the two libraries the same machine code" for the library as a whole; its per-kernel byte hashes
are not comparable across trees (kernels address device globals PC-relatively: a change anywhere in a code object moves literals of
kernels whose source did not change).  This tool compares INSTRUCTION STREAMS: every gfx950 code object is disassembled
(llvm-objdump), every kernel's instructions are taken as text -- mnemonics and operands, no addresses, no encodings; branch offsets
are relative and stay -- and the literal of the `s_add_u32 / s_addc_u32` pair that follows an `s_getpc_b64` (the PC-relative address
of a device global or of another function) is masked.  Two kernels with the same normalised stream and the same kernel descriptor
(register counts, LDS size, ...) execute the same instructions; measurements of one hold for the other.

    python tools/isa_identity.py <old libjutul_hip.so> [<new libjutul_hip.so> = the in-tree one]  [--all]
prints the kernels that differ (with the number of differing instructions) and a summary; --all lists the identical ones too;
--regions: for every differing kernel the window of instructions that holds all differences other than branch displacements.
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fatbin_hash import code_objects, fatbin_sections, kernel_hashes  # noqa: E402

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def kernel_streams(lib):
    """{mangled kernel name: [normalised instruction, ...]} over all gfx950 code objects of the library"""
    kernels = set(kernel_hashes(lib))          # functions that have a kernel descriptor
    out = {}
    with tempfile.TemporaryDirectory() as td:
        n = 0
        for name, blob in fatbin_sections(lib):
            if name != ".hip_fatbin":
                continue
            for _triple, obj in code_objects(blob):
                p = os.path.join(td, f"co{n}.elf")
                n += 1
                with open(p, "wb") as f:
                    f.write(obj)
                txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", p], capture_output=True, text=True, check=True).stdout
                cur, pc_window = None, 0
                for line in txt.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                    if m:
                        cur = m.group(1) if m.group(1) in kernels else None
                        if cur is not None:
                            out[cur] = []
                        continue
                    if cur is None or not line.startswith("\t"):
                        continue
                    ins = line.split("//")[0].strip()
                    ins = re.sub(r"\s+", " ", ins)
                    if not ins or ins.startswith("s_nop") or ins.startswith("s_code_end"):
                        continue                # (alignment padding behind s_endpgm)
                    if ins.startswith("s_getpc_b64"):
                        pc_window = 4
                    elif pc_window > 0:
                        pc_window -= 1
                        if re.match(r"s_addc?_u32 ", ins):
                            ins = re.sub(r"(0x[0-9a-f]+|-?\d+)$", "<pcrel>", ins)
                    out[cur].append(ins)
    return out


def descriptors(lib):
    """{kernel: sha of its 64-byte kernel descriptor}: registers, LDS, scratch, enabled SGPR inputs"""
    import struct
    out = {}
    for name, blob in fatbin_sections(lib):
        if name != ".hip_fatbin":
            continue
        for _triple, obj in code_objects(blob):
            shoff = struct.unpack_from("<Q", obj, 0x28)[0]
            shentsize, shnum, _ = struct.unpack_from("<HHH", obj, 0x3A)
            secs = [struct.unpack_from("<IIQQQQIIQQ", obj, shoff + i * shentsize) for i in range(shnum)]
            for _n, stype, _f, _a, off, size, link, _i, _al, entsize in secs:
                if stype != 2:
                    continue
                stroff = secs[link][4]
                for k in range(size // entsize):
                    st_name, _info, _o, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", obj, off + k * entsize)
                    if st_shndx == 0 or st_shndx >= len(secs) or st_size != 64:
                        continue
                    end = obj.index(b"\0", stroff + st_name)
                    nm = obj[stroff + st_name:end].decode()
                    if not nm.endswith(".kd"):
                        continue
                    sec = secs[st_shndx]
                    kd = bytearray(obj[sec[4] + st_value - sec[3]:sec[4] + st_value - sec[3] + 64])
                    kd[16:24] = b"\0" * 8          # kernel_code_entry_byte_offset: where the code lies relative to the descriptor
                    out[nm[:-3]] = hashlib.sha256(bytes(kd)).hexdigest()[:16]
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:  # noqa: BLE001
        return {n: n for n in names}


def compare(old, new):
    so, sn = kernel_streams(old), kernel_streams(new)
    do, dn = descriptors(old), descriptors(new)
    rows = []
    for k in sorted(set(so) | set(sn)):
        if k not in so:
            rows.append((k, "only in new", len(sn[k]), 0)); continue
        if k not in sn:
            rows.append((k, "only in old", len(so[k]), 0)); continue
        a, b = so[k], sn[k]
        if a == b and do.get(k) == dn.get(k):
            rows.append((k, "identical", len(a), 0)); continue
        import difflib
        sm = difflib.SequenceMatcher(a=a, b=b, autojunk=False)
        changed = sum(max(i2 - i1, j2 - j1) for tag, i1, i2, j1, j2 in sm.get_opcodes() if tag != "equal")
        rows.append((k, "differs" if a != b else "same instructions, other descriptor", len(b), changed))
    return rows


BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch) -?\d+$")


def changed_window(a, b):
    """(first, last, changed, hunks, sleeps) of the instructions of b that differ from a, NOT counting branches whose only difference
    is their displacement (a branch across a region that grew or shrank); sleeps = s_sleep instructions inside the window (the
    cross-rank wait loop is the only place these kernels sleep)"""
    import difflib
    sm = difflib.SequenceMatcher(a=a, b=b, autojunk=False)
    hunks = []
    for tag, i1, i2, j1, j2 in sm.get_opcodes():
        if tag == "equal":
            continue
        if tag == "replace" and i2 - i1 == j2 - j1 and all(BRANCH.match(x) and BRANCH.match(y) and x.split()[0] == y.split()[0]
                                                            for x, y in zip(a[i1:i2], b[j1:j2])):
            continue
        hunks.append((j1, j2, i1, i2))
    if not hunks:
        return None
    lo, hi = min(h[0] for h in hunks), max(h[1] for h in hunks)
    return lo, hi, sum(max(h[1] - h[0], h[3] - h[2]) for h in hunks), len(hunks), sum(1 for x in b[lo:hi] if x.startswith("s_sleep"))


def regions(old, new):
    so, sn = kernel_streams(old), kernel_streams(new)
    names = demangle(sorted(sn))
    print("# kernels whose instruction streams differ: window [first, last) of the new kernel that holds every difference other than branch")
    print("# displacements, instructions changed inside it, hunks, s_sleep instructions inside it (= the cross-rank wait loop)")
    for k in sorted(sn):
        if k not in so or so[k] == sn[k]:
            continue
        w = changed_window(so[k], sn[k])
        if w is None:
            print(f"only branch displacements                                   {len(sn[k]):6d} instructions   {names[k][:120]}")
        else:
            print(f"window [{w[0]:5d},{w[1]:5d}) {w[2]:4d} changed in {w[3]:3d} hunks, {w[4]} s_sleep  {len(sn[k]):6d} instructions   {names[k][:120]}")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    old = args[0]
    new = args[1] if len(args) > 1 else os.path.join(root, "jutul.jl_amd", "libjutul_hip.so")
    if "--regions" in sys.argv:
        regions(old, new)
        return
    rows = compare(old, new)
    names = demangle([r[0] for r in rows])
    same = [r for r in rows if r[1] == "identical"]
    print(f"# {len(rows)} kernels: {len(same)} with identical instruction streams and descriptors, {len(rows) - len(same)} not")
    for k, what, n, changed in rows:
        if what == "identical" and "--all" not in sys.argv:
            continue
        print(f"{what:38s} {n:6d} instructions  {changed:5d} changed   {names[k][:150]}")


if __name__ == "__main__":
    main()
