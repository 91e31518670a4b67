#!/usr/bin/env python3
"""VGPRs / scratch / spills of the kernels in a built library (no GPU needed): scratch/kernel_resources.py [lib.so] [regex]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "quadruped-trajectory-optimization-stack_amd", "csrc", "libqtos_planner.so")
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_step|k_start|k_chordILi(96|112)E|k_kkt\dILi(96|112)")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
    sizes = subprocess.check_output([LLVM + "llvm-readelf", "-s", "-W", co], text=True)
size = {ln.split()[7]: int(ln.split()[2]) for ln in sizes.splitlines() if len(ln.split()) == 8 and ln.split()[3] == "FUNC"}
for k in notes.split("- .agpr_count")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", k).group(1)
    if not pat.search(nm):
        continue
    g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, k).group(1))
    print("%-60s vgpr %3d  scratch %4d B  sgpr spills %3d  vgpr spills %3d  code %6d B" %
          (subprocess.check_output(["c++filt", nm], text=True).strip()[:60], g("vgpr_count"), g("private_segment_fixed_size"),
           g("sgpr_spill_count"), g("vgpr_spill_count"), size.get(nm, 0)))
