"""Build-time gate on the one-XCD kernels: k_cg_xcd / k_keff_xcd hand p, r, x, q, raw, phi, tf and the block partials
from workgroup to workgroup INSIDE one launch, behind a counter barrier among the workgroups that share one XCD's L2.  The readers
are only correct if those loads miss the compute unit's L1, i.e. if every one of them still carries the non-temporal hint in the
machine code -- and the optimiser has dropped that hint once already (two loads of one address in the arms of a branch were merged into
a plain one, nf_kernels.h "the streaming-load hint never reached x").  An agent-scope acquire after each barrier would make plain loads
legal, but it invalidates / bypasses the XCD's L2 as well (measured: 8.9 us per barrier round against 0.9).

So the hint is checked where it matters: this test takes the gfx950 code object out of the built libneutfem_hip.so, disassembles it and
counts, per instantiation, the global loads WITHOUT nt / sc0 / sc1.  What may legitimately be plain are the loads of data no workgroup
writes during the launch (line factors of the overlap cell, 1/d of a line's first face, Chebyshev tables, the D / geometry of bubble
constants): their number per instantiation is pinned below.  One exchanged-vector load losing its hint raises the count and fails the
suite; so does a new plain load that nobody classified.  No GPU needed (the compiler of the GPU box is this image's)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "neutfem_amd", "lib", "libneutfem_hip.so")

# plain (hint-less) global loads per instantiation <NCH, VEC, NB, SEG>, by kernel and NB, as built from the reviewed sources of round 4
PLAIN_MAX = {("k_cg_xcd", 0): 22, ("k_cg_xcd", 1): 70, ("k_cg_xcd", 2): 72, ("k_keff_xcd", 0): 12, ("k_keff_xcd", 1): 60, ("k_keff_xcd", 2): 62}


def _disassemble(tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", LIB, os.path.join(tmp, "discard.so")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}"])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def test_exchanged_vector_loads_of_the_xcd_kernels_keep_their_cache_bypass(tmp_path):
    if not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "clang-offload-bundler")) and shutil.which("c++filt")):
        pytest.skip("llvm-objdump / clang-offload-bundler / c++filt not in this image")
    assert os.path.exists(LIB), "libneutfem_hip.so is missing: run __graft_entry__.build()"
    asm = _disassemble(str(tmp_path))
    seen = {}
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", asm):
        m = re.match(r"[0-9a-f]{16} <(_ZN2nf\d+k_(?:cg|keff)_xcd\S+)>:", f)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        t = re.search(r"nf::(k_\w+_xcd)<(\d+), (true|false), (\d+), (\d+)>", name)
        assert t, name
        loads = [l for l in f.splitlines() if re.search(r"\bglobal_load", l)]
        plain = [l for l in loads if not re.search(r"\b(nt|sc0|sc1)\b", l.split("//")[0])]
        nt = [l for l in loads if re.search(r"\bnt\b", l.split("//")[0])]
        key = (t.group(1), int(t.group(4)))
        seen[t.group(0)] = (len(loads), len(plain))
        assert len(nt) >= 40, (name, len(loads), len(nt), len(plain))                   # the exchanged vectors ARE read with the hint (the sc1 loads are the barrier's own polls)
        assert len(plain) <= PLAIN_MAX[key], f"{name}: {len(plain)} global loads without nt/sc0/sc1 (reviewed maximum {PLAIN_MAX[key]}): a load of a vector that " \
                                             f"other workgroups write during the launch may have lost its L1 bypass -- read the disassembly before raising the bound"
    assert len(seen) == 24, sorted(seen)                             # 12 instantiations of each kernel: none went missing from the check
