#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use of the gfx950 code objects inside libgoblin_hip.so.

    python tools/kernel_resources.py [--spills] [--all] [--json] [library]

The library is a host ELF whose .hip_fatbin section bundles one device ELF per translation unit; this scans for
the embedded AMDGPU ELF images, runs llvm-readelf --notes on each and prints the kernel descriptors' metadata.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"
EM_AMDGPU = 224


def device_images(blob):
    pos = 0
    while True:
        pos = blob.find(b"\x7fELF\x02\x01\x01", pos)
        if pos < 0:
            return
        hdr = blob[pos:pos + 64]
        if len(hdr) == 64 and struct.unpack_from("<H", hdr, 18)[0] == EM_AMDGPU:
            shoff, = struct.unpack_from("<Q", hdr, 40)
            shentsize, shnum = struct.unpack_from("<HH", hdr, 58)
            size = shoff + shentsize * shnum
            yield blob[pos:pos + size]
            pos += max(size, 1)
        else:
            pos += 4


def kernels_of(image):
    import yaml
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    a = text.find("---")
    b = text.find("\n...", a)
    if a < 0:
        return []
    meta = yaml.safe_load(text[a + 3:b if b > 0 else None])
    out = []
    for k in meta.get("amdhsa.kernels", []):
        out.append({key.lstrip("."): val for key, val in k.items() if key != ".args"})
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "goblin_amd", "lib", "libgoblin_hip.so")
    blob = open(lib, "rb").read()
    rows = []
    for img in device_images(blob):
        rows += kernels_of(img)
    names = subprocess.run([FILT], input="\n".join(r["symbol"].replace(".kd", "") for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = n
    if "--all" not in sys.argv:   # hipCUB's radix sort (device BVH build) brings ~200 library kernels
        rows = [r for r in rows if "rocprim" not in r["name"]]
    if "--spills" in sys.argv:
        rows = [r for r in rows if r.get("vgpr_spill_count", 0) or r.get("sgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)]
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
        return
    print("%5s %5s %5s %6s %6s %8s %7s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratchB", "ldsB", "kernel"))
    for r in sorted(rows, key=lambda r: r["name"]):
        print("%5d %5d %5d %6d %6d %8d %7d  %s" % (r.get("vgpr_count", -1), r.get("agpr_count", 0), r.get("sgpr_count", -1), r.get("vgpr_spill_count", 0),
                                                    r.get("sgpr_spill_count", 0), r.get("private_segment_fixed_size", 0),
                                                    r.get("group_segment_fixed_size", 0), r["name"][:150]))
    print("%d kernels" % len(rows))


if __name__ == "__main__":
    main()
