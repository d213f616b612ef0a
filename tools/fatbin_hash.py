#!/usr/bin/env python3
"""sha256 of the DEVICE code of libjutul_hip.so: the machine code (.text) and the kernel descriptors / constants (.rodata) of
every gfx950 code object in its .hip_fatbin section.  (The section as a whole is not comparable between two builds: every
compilation unit carries a `__hip_cuid_<hash>` symbol derived from its source path and text, which also reorders the symbol
tables.)  Two trees with the same hash run the same kernels -- host-side changes (set-up, bindings, options) do not move it -- so
measurements committed under profiles/ for one of them hold for the other.

    python tools/fatbin_hash.py [path/to/libjutul_hip.so]            one hash for the whole library
    python tools/fatbin_hash.py --kernels [path/to/libjutul_hip.so]  one hash per kernel (mangled names) -- NOT a per-kernel
        identity test across different trees: kernels address device globals and each other PC-relatively, so a change anywhere in a
        code object moves the literals of kernels whose source did not change (tried on 3b0db32 vs d92f7eb: 171 of 317 differ)
"""
import hashlib
import os
import struct
import sys


def fatbin_sections(path):
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"\x7fELF" and data[4] == 2, "64-bit ELF expected"
    shoff = struct.unpack_from("<Q", data, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    stroff = secs[shstrndx][4]
    out = []
    for name_off, _type, _flags, _addr, off, size, *_ in secs:
        end = data.index(b"\0", stroff + name_off)
        name = data[stroff + name_off:end].decode()
        if name in (".hip_fatbin", ".hipFatBinSegment"):
            out.append((name, data[off:off + size]))
    return out


def elf_sections(data):
    shoff = struct.unpack_from("<Q", data, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    stroff = secs[shstrndx][4]
    out = {}
    for name_off, stype, _flags, _addr, off, size, *_ in secs:
        end = data.index(b"\0", stroff + name_off)
        if stype != 8:  # (SHT_NOBITS has no bytes)
            out[data[stroff + name_off:end].decode()] = data[off:off + size]
    return out


def code_objects(fatbin):
    """the gfx950 ELF code objects of all clang offload bundles in the section (one bundle per compilation unit)"""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        pos = fatbin.find(magic, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", fatbin, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fatbin, q)
            triple = fatbin[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, fatbin[pos + off:pos + off + size]
        pos += 24


def kernel_hashes(path):
    """{mangled kernel name: sha256[:16] of its machine code + kernel descriptor} over all code objects"""
    out = {}
    for name, blob in fatbin_sections(path):
        if name != ".hip_fatbin":
            continue
        for _triple, obj in code_objects(blob):
            shoff = struct.unpack_from("<Q", obj, 0x28)[0]
            shentsize, shnum, _ = struct.unpack_from("<HHH", obj, 0x3A)
            secs = [struct.unpack_from("<IIQQQQIIQQ", obj, shoff + i * shentsize) for i in range(shnum)]
            for _n, stype, _f, _a, off, size, link, _i, _al, entsize in secs:
                if stype != 2:  # SHT_SYMTAB
                    continue
                stroff = secs[link][4]
                syms = {}
                for k in range(size // entsize):
                    st_name, st_info, _o, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", obj, off + k * entsize)
                    if st_shndx == 0 or st_shndx >= len(secs) or st_size == 0:
                        continue
                    end = obj.index(b"\0", stroff + st_name)
                    nm = obj[stroff + st_name:end].decode()
                    sec = secs[st_shndx]
                    syms[nm] = obj[sec[4] + st_value - sec[3]:sec[4] + st_value - sec[3] + st_size]
                for nm, code in syms.items():
                    if nm.endswith(".kd") or nm + ".kd" not in syms:
                        continue  # kernels only: functions that have a kernel descriptor
                    out[nm] = hashlib.sha256(code + syms[nm + ".kd"]).hexdigest()[:16]
    return out


def device_code_hash(path):
    h = hashlib.sha256()
    n_obj = n_text = 0
    for name, blob in fatbin_sections(path):
        if name != ".hip_fatbin":
            continue
        for triple, obj in code_objects(blob):
            secs = elf_sections(obj)
            h.update(triple.encode())
            for sec in (".text", ".rodata"):
                h.update(sec.encode() + struct.pack("<Q", len(secs.get(sec, b""))) + secs.get(sec, b""))
            n_obj += 1
            n_text += len(secs.get(".text", b""))
    return h.hexdigest()[:16], n_obj, n_text


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(root, "jutul.jl_amd", "libjutul_hip.so")
    if "--kernels" in sys.argv:
        path = [a for a in sys.argv[1:] if not a.startswith("--")]
        path = path[0] if path else os.path.join(root, "jutul.jl_amd", "libjutul_hip.so")
        for nm, hh in sorted(kernel_hashes(path).items()):
            print(hh, nm)
        return
    digest, n_obj, n_text = device_code_hash(path)
    print(f"{digest}  {n_obj} gfx950 code objects, {n_text} bytes of machine code  {path}")


if __name__ == "__main__":
    main()
