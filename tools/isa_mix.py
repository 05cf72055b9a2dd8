#!/usr/bin/env python3
"""Share of v_mad_u64_u32 among the VALU instructions of every kernel's shipped machine code (static count over the disassembly
of the gfx950 code objects in bulletproofs_amd/csrc/build/*.o).  The hot kernels are straight-line field arithmetic inside short
loops, so the static share is the executed one to within a per cent; bench.py weights it by each kernel's SQ_INSTS_VALU.
    python tools/isa_mix.py [objects...]   ->  JSON {kernel: {valu, mad_u64, fraction}}"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402


def mix_of(path):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        co = kr._code_object(path, tmp)
        if co is None:
            return out
        dis = subprocess.run([kr.LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    cur = None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            out[cur] = [0, 0]
            continue
        if cur is None:
            continue
        f = ln.split()
        if f and f[0].startswith("v_"):
            out[cur][0] += 1
            if f[0] == "v_mad_u64_u32":
                out[cur][1] += 1
    names = [k for k in out if not re.match(r"^L\d+$", k)]
    dm = kr._demangle(names)
    res = {}
    for k in names:
        v, m_ = out[k]
        # local labels (<L12>) belong to the kernel before them: objdump --symbolize-operands is off, so there are none here
        if v:
            res[dm.get(k, k)] = {"valu": v, "mad_u64": m_, "fraction": round(m_ / v, 4)}
    return res


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "bulletproofs_amd", "csrc", "build", "*.o")))
    allk = {}
    for o in objs:
        allk.update(mix_of(o))
    print(json.dumps(allk, indent=1))
