#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures of the gfx950 code objects inside the built objects (no GPU needed):

    python tools/kernel_resources.py [csv-out]

For every csrc/*.o: the .hip_fatbin section is unbundled (clang-offload-bundler) and the AMDGPU metadata note of the code
object read (llvm-readelf --notes): .vgpr_count, .agpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count,
.private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size (static LDS).  What profiles/README.md quotes.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
        ".private_segment_fixed_size", ".group_segment_fixed_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)\(", n)
    return m.group(1) if m else n[:80]


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        if os.path.getsize(fat) == 0:
            return []
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co, "--unbundle"])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    ks, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == ".agpr_count":          # first key of a kernel's map in the note (alphabetical)
            cur = {}
            ks.append(cur)
        if cur is not None and (k in KEYS or k == ".name"):
            cur[k] = v.strip("'\"")
    ks = [k for k in ks if ".name" in k]
    for k, dn in zip(ks, demangle([k[".name"] for k in ks])):
        k["kernel"] = short(dn)
    return ks


def main():
    rows = []
    for obj in sorted(glob.glob(os.path.join(REPO, "fly_bproject_amd", "csrc", "*.o"))):
        for k in kernels_of(obj):
            rows.append((os.path.basename(obj)[:-2], k))
    hdr = ["file", "kernel", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes_per_lane", "static_lds_bytes"]
    lines = [",".join(hdr)]
    for f, k in rows:
        lines.append(",".join([f, '"%s"' % k["kernel"]] + [str(k.get(x, "")) for x in KEYS]))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
