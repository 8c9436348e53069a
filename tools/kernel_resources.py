#!/usr/bin/env python3
"""Per-kernel resources (VGPRs, scratch bytes per lane = spills, LDS) of every gfx950 kernel in a built libfedfr_hip.so, read from the code
objects' AMDGPU metadata — no GPU, no ROCm tool: the clang offload bundles inside the .hip_fatbin section are parsed by hand (the image's
roc-obj-ls needs a perl module it does not have).  usage: python tools/kernel_resources.py [lib.so] [name-filter]

tests/test_abi.py::test_hot_kernels_do_not_spill uses kernels(): an edit to a shared kernel template can push an instantiation that sits
at the register limit into scratch without any test failing (round 3: a new epilogue branch in conv_glds_impl.h took the two-tiles
28x28 conv from 254 VGPRs to 256 + 128 B of scratch per lane, 34 -> 51 us per launch)."""
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundles(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            yield triple, blob[i + off: i + off + size]
        pos = i + 24


def _notes(elf):
    """NT_AMDGPU_METADATA (type 32, owner AMDGPU) msgpack blobs of an ELF64 little-endian code object."""
    if elf[:4] != b"\x7fELF":
        return
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type = struct.unpack_from("<I", elf, sh + 4)[0]
        if sh_type != 7:          # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12: p + 12 + namesz]
            d0 = p + 12 + (namesz + 3) // 4 * 4
            if ntype == 32 and name.startswith(b"AMDGPU"):
                yield elf[d0: d0 + descsz]
            p = d0 + (descsz + 3) // 4 * 4


def kernels(path):
    """{kernel symbol: {"vgpr": n, "agpr": n, "sgpr": n, "scratch": bytes per lane, "lds": static bytes}} over all gfx950 code objects."""
    blob = open(path, "rb").read()
    out = {}
    for triple, obj in _bundles(blob):
        if "gfx950" not in triple:
            continue
        for note in _notes(obj):
            md = msgpack.unpackb(note, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                out[k[".name"]] = {"vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0),
                                   "scratch": k.get(".private_segment_fixed_size", 0), "lds": k.get(".group_segment_fixed_size", 0)}
    return out


if __name__ == "__main__":
    import os
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fedfr_amd", "libfedfr_hip.so")
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = kernels(lib)
    for name in sorted(ks, key=lambda n: (-ks[n]["scratch"], -ks[n]["vgpr"])):
        if filt in name:
            r = ks[name]
            print("%-110s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d" % (name[:110], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"]))
    print("%d kernels, %d with scratch" % (len(ks), sum(1 for r in ks.values() if r["scratch"])))
