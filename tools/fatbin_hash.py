#!/usr/bin/env python3
"""sha256 of the DEVICE code of libjutul_hip.so: the machine code (.text) and the kernel descriptors / constants (.rodata) of
every gfx950 code object in its .hip_fatbin section.  (The section as a whole is not comparable between two builds: every
compilation unit carries a `__hip_cuid_<hash>` symbol derived from its source path and text, which also reorders the symbol
tables.)  Two trees with the same hash run the same kernels -- host-side changes (set-up, bindings, options) do not move it -- so
measurements committed under profiles/ for one of them hold for the other.

    python tools/fatbin_hash.py [path/to/libjutul_hip.so]
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
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "jutul.jl_amd", "libjutul_hip.so")
    digest, n_obj, n_text = device_code_hash(path)
    print(f"{digest}  {n_obj} gfx950 code objects, {n_text} bytes of machine code  {path}")


if __name__ == "__main__":
    main()
