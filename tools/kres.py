"""Kernel resource table from hipcc -S output (/tmp/asm/*.s): scratch bytes per lane, static LDS, VGPRs, spills."""
import glob
import re
import subprocess
import sys

def demangle(n):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            return subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
        except FileNotFoundError:
            continue
    return n

for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "/tmp/asm") + "/*.s")):
    txt = open(f).read()
    for k in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
        g = lambda key: re.search(r"\." + key + r":\s+(\S+)", k).group(1)
        print(f"{f.split('/')[-1][:-2]:17s} scratch={g('private_segment_fixed_size'):>6s} lds={g('group_segment_fixed_size'):>6s} "
              f"vgpr={g('vgpr_count'):>4s} agpr={g('agpr_count'):>3s} spill={g('vgpr_spill_count'):>4s} {demangle(g('name'))[:100]}")
