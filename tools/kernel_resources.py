#!/usr/bin/env python3
"""VGPRs / spills / scratch / LDS of every kernel in a built library or object (from the code objects' metadata notes).
usage: python tools/kernel_resources.py mocodad_amd/libmocodad_hip.so [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
path, flt = os.path.abspath(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as td:
    tmp = os.path.join(td, os.path.basename(path))
    os.symlink(path, tmp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
    for f in sorted(os.listdir(td)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, f)], stdout=subprocess.PIPE, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run([f"{LLVM}/llvm-cxxfilt", g("name")], stdout=subprocess.PIPE, text=True).stdout.strip() if os.path.exists(f"{LLVM}/llvm-cxxfilt") else g("name")
            if flt in name:
                print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} (spill {g('vgpr_spill_count')}) sgpr {g('sgpr_count'):>4s} (spill {g('sgpr_spill_count')}) scratch {g('private_segment_fixed_size'):>5s} B  lds {g('group_segment_fixed_size')}")
