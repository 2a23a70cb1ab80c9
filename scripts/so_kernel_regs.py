#!/usr/bin/env python3
"""Register / LDS / spill table of the kernels inside the built libtangram_hip.so (code-object metadata): so_kernel_regs.py [filter ...]"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "tangram_amd", "csrc", "libtangram_hip.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat")
    subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so])
    b = open(fat, "rb").read()
    cnt = struct.unpack("<Q", b[24:32])[0]; o = 32
    for _ in range(cnt):
        off, size, tl = struct.unpack("<QQQ", b[o:o + 24]); o += 24; t = b[o:o + tl].decode(); o += tl
        if "gfx950" in t:
            open(os.path.join(d, "co"), "wb").write(b[off:off + size])
    md = subprocess.run([f"{llvm}/llvm-readelf", "--notes", os.path.join(d, "co")], capture_output=True, text=True).stdout
names, rows = [], []
for it in re.split(r"\n  - ", md[md.find("amdhsa.kernels"):])[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, it) or [None, "?"])[1]
    name = re.search(r"\.name:\s+(\S+)", it)
    if name:
        names.append(name.group(1)); rows.append((g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print("vgpr agpr sgpr spill scratch  kernel")
for r, n in zip(rows, dem):
    if not sys.argv[1:] or any(f in n for f in sys.argv[1:]):
        print("%4s %4s %4s %5s %7s  %s" % (*r, n[:150]))
