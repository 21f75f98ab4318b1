"""Registers, scratch and LDS of every kernel in a .hip file (cross-compiled here): python tools/kres.py hept_amd/csrc/block_attn.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
asm = "/tmp/kres_%d.s" % os.getpid()
flags = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1".split()
subprocess.run(["/opt/rocm/bin/hipcc", *flags, *sys.argv[3:], "-S", "--cuda-device-only", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
os.remove(asm)
names = re.findall(r"\.name:\s+(\S+)", s)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
blocks = s.split("  - .agpr_count:")[1:]
for b in blocks:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    n = g("name")
    d = dem[names.index(n)].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if flt in d:
        print(f"{d:64s} vgpr={g('vgpr_count'):>3s} sgpr={g('sgpr_count'):>3s} scratch={g('private_segment_fixed_size'):>4s} lds={g('group_segment_fixed_size'):>6s} spill={g('vgpr_spill_count')}")
