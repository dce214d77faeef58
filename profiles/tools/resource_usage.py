#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel whose demangled name matches a pattern, from hipcc's own report
(-Rpass-analysis=kernel-resource-usage on the device compile; no GPU needed).
Usage: python profiles/tools/resource_usage.py [regex on the demangled name]   (default: every kernel with scratch)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "neutfem_amd", "csrc", "neutfem_hip.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                    "-c", "-o", "/dev/null", src], capture_output=True, text=True)
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
for b in re.split(r"remark: Function Name: ", r.stderr)[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void nf::", "", dem); dem = dem.split("(")[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    v, a, sp, occ, lds = g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    if (pat and pat.search(dem)) or (not pat and sp):
        print(f"{dem:110s} VGPR {v:3d} AGPR {a:3d} scratch {sp:4d} B/lane  occupancy {occ}  static LDS {lds}")
