"""Diagnostic: registers / LDS / scratch of device kernels in an object file (names containing any of the given substrings).
usage: python tools/kres.py treeqp_amd/lib_var/dev/tdunes_device.hip.o k_hf_w k_sg"""
import re, subprocess, sys, glob, os
obj = sys.argv[1]
pats = sys.argv[2:]
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", obj], check=True, capture_output=True)
co = glob.glob(obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950")[0]
t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for f in glob.glob(obj + ".0.*"):
    os.remove(f)
for b in re.split(r"\n\s+- \.agpr_count", t)[1:]:
    m = re.search(r"\.name:\s+(\S+)", b)
    if not m or not any(p in m.group(1) for p in pats):
        continue
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
    print(f"{m.group(1)[:70]:70s} vgpr {g('vgpr_count'):3d} agpr {int(b.split()[1]) if b.split()[0]==':' else b.split()[0]} sgpr {g('sgpr_count'):3d} spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')} lds {g('group_segment_fixed_size')} scratch {g('private_segment_fixed_size')}")
