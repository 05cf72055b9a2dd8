#!/usr/bin/env python3
"""Register, scratch and code-size figures of every gfx950 kernel in the given objects, read from the code object's own
metadata (llvm-readelf --notes) and symbol table -- what the hardware will be given, not what the source suggests.
    python tools/kernel_resources.py bulletproofs_amd/csrc/build/*.o        (host objects with an embedded .hip_fatbin)
    python tools/kernel_resources.py some.co                                (a bare code object: hipcc --cuda-device-only)
`check(objs, allow)` is what __graft_entry__.build() calls: a kernel whose private segment (scratch) or VGPR spill count is
non-zero must be on the allow-list, with the figures it is allowed -- a hot-path kernel that silently starts spilling fails
the build."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def _code_object(path, tmp):
    with open(path, "rb") as f:
        head = f.read(4)
    if head == b"\x7fELF":
        out = subprocess.run([LLVM + "/llvm-readelf", "-h", path], capture_output=True, text=True).stdout
        if "amdgpu" in out.lower() or "amd gpu" in out.lower():
            return path
    with open(path, "rb") as f:
        bundle = f.read(24) == b"__CLANG_OFFLOAD_BUNDLE__"
    fat = os.path.join(tmp, os.path.basename(path) + ".fat")
    if bundle:   # hipcc --cuda-device-only -c
        fat = path
    else:
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return None
    co = os.path.join(tmp, os.path.basename(path) + ".co")
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def _demangle(names):
    if not names:
        return {}
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return {n: re.sub(r"\(.*", "", d) for n, d in zip(names, out)}


def kernels_of(path):
    """[{name, vgpr, agpr, sgpr, scratch, spill_vgpr, spill_sgpr, code_bytes}] for one object file"""
    with tempfile.TemporaryDirectory() as tmp:
        co = _code_object(path, tmp)
        if co is None:
            return []
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", co], capture_output=True, text=True).stdout
    size = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    ks, cur = [], None
    for ln in notes.splitlines():
        ln = ln.strip()
        if ln.startswith("- .") or ln.startswith("- "):
            if cur and "name" in cur:
                ks.append(cur)
            cur = {}
            ln = ln[2:]
        if cur is None:
            continue
        m = re.match(r"\.(agpr_count|vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|name):\s+(\S+)", ln)
        if m:
            cur[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
    if cur and "name" in cur:
        ks.append(cur)
    ks = [k for k in ks if "vgpr_count" in k]
    dm = _demangle([k["name"] for k in ks])
    return [dict(name=dm.get(k["name"], k["name"]), vgpr=k.get("vgpr_count", 0), agpr=k.get("agpr_count", 0), sgpr=k.get("sgpr_count", 0),
                 scratch=k.get("private_segment_fixed_size", 0), spill_vgpr=k.get("vgpr_spill_count", 0), spill_sgpr=k.get("sgpr_spill_count", 0),
                 code_bytes=size.get(k["name"], 0)) for k in ks]


def report(paths):
    rows = []
    for p in paths:
        for k in kernels_of(p):
            rows.append((os.path.basename(p), k))
    print("%-14s %-46s %5s %5s %5s %8s %6s %9s" % ("object", "kernel", "vgpr", "agpr", "sgpr", "scratch", "spills", "code"))
    for obj, k in rows:
        print("%-14s %-46s %5d %5d %5d %8d %6d %9d" % (obj, k["name"][:46], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["spill_vgpr"], k["code_bytes"]))
    return rows


def check(paths, allow):
    """allow: {kernel name prefix: (max scratch bytes, max spilled VGPRs)}.  Returns a list of violation strings."""
    bad = []
    for p in paths:
        for k in kernels_of(p):
            if k["scratch"] == 0 and k["spill_vgpr"] == 0:
                continue
            lim = None
            for pre, v in allow.items():
                if k["name"].startswith(pre) and (lim is None or len(pre) > lim[0]):
                    lim = (len(pre), v)
            if lim is None:
                bad.append("%s: %d B of scratch, %d spilled VGPRs and no allow-list entry" % (k["name"], k["scratch"], k["spill_vgpr"]))
            elif k["scratch"] > lim[1][0] or k["spill_vgpr"] > lim[1][1]:
                bad.append("%s: %d B of scratch / %d spilled VGPRs exceed the allowed %d / %d" % (k["name"], k["scratch"], k["spill_vgpr"], lim[1][0], lim[1][1]))
    return bad


if __name__ == "__main__":
    report(sys.argv[1:])
