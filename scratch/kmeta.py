#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every step kernel in a built libcagpu*.so (the code object's own notes).
usage: kmeta.py [path/to/libcagpu.so] [substring of the demangled kernel name]"""
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                         "gym_collision_avoidance_amd", "libcagpu.so")
pat = sys.argv[2] if len(sys.argv) > 2 else "ca_"
data = open(lib, "rb").read()
tmp = tempfile.mkdtemp()
co = os.path.join(tmp, "co.o")
for m in re.finditer(b"\x7fELF", data):
    i = m.start()
    if struct.unpack_from("<H", data, i + 18)[0] == 224:  # EM_AMDGPU
        shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
        shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
        open(co, "wb").write(data[i:i + shoff + shentsize * shnum])
        break
else:
    sys.exit("no device code object in %s" % lib)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    dn = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
    if pat not in dn:
        continue
    g = lambda f: re.search(r"\.%s:\s+(\d+)" % f, k).group(1)
    print("%-84s vgpr %3s sgpr %3s spill v%-3s s%-3s scratch %4s lds %6s" % (
        re.sub(r"\(anonymous namespace\)::", "", dn)[:84], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
        g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
shutil.rmtree(tmp)
