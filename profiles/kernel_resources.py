#!/usr/bin/env python3
"""Per-kernel resources of the built library, read from the gfx950 code objects' metadata (no GPU needed): registers, spills, scratch (private segment) and static LDS per kernel.
usage: python profiles/kernel_resources.py [jm_amd/libjmhip.so] > profiles/rNN_kernel_resources.txt"""
import os
import re
import struct
import subprocess
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "jm_amd", "libjmhip.so")
rows = []
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    d = open(fat, "rb").read()
    for n, m in enumerate(re.finditer(b"\x7fELF", d)):       # every translation unit's code object: an ELF64 image inside the offload bundle
        i = m.start()
        e_shoff = struct.unpack_from("<Q", d, i + 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from("<HH", d, i + 0x3A)
        p = os.path.join(tmp, f"co{n}.elf")
        open(p, "wb").write(d[i:i + e_shoff + e_shentsize * e_shnum])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", p], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode(errors="replace")
        cur = {}
        for line in notes.splitlines():
            mm = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
            if not mm:
                continue
            k, v = mm.groups()
            if k == "agpr_count" and cur.get("name"):
                rows.append(cur); cur = {}
            if k in ("agpr_count", "group_segment_fixed_size", "name", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size"):
                cur[k] = v.strip()
        if cur.get("name"):
            rows.append(cur)
def short(n):                                                # _Z<len><name>...: the function's name without its signature (templates keep their I...E suffix)
    m = re.match(r"_Z(\d+)", n)
    if not m:
        return n
    k = int(m.group(1))
    name = n[m.end():m.end() + k]
    rest = n[m.end() + k:]
    t = re.match(r"I((?:Lb[01]E)+)E", rest)                  # bool template arguments, as mb_pipe's <EPZS, T8>
    return name + ("<" + ",".join(re.findall(r"Lb([01])E", t.group(1))) + ">" if t else "")
print(f"# {os.path.relpath(so, ROOT)}: kernels of the gfx950 code objects (metadata notes) -- vgpr / agpr / sgpr, spilled vgpr / sgpr, scratch bytes per lane, static LDS bytes, max workgroup")
print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>7s} {'sspill':>7s} {'scratch':>8s} {'lds':>7s} {'wg':>5s}")
for r in sorted(rows, key=lambda r: short(r["name"])):
    print(f"{short(r['name'])[:58]:58s} {r.get('vgpr_count', '?'):>5s} {r.get('agpr_count', '?'):>5s} {r.get('sgpr_count', '?'):>5s} {r.get('vgpr_spill_count', '?'):>7s} {r.get('sgpr_spill_count', '?'):>7s} "
          f"{r.get('private_segment_fixed_size', '?'):>8s} {r.get('group_segment_fixed_size', '?'):>7s} {r.get('max_flat_workgroup_size', '?'):>5s}")
