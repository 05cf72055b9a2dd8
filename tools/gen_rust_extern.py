#!/usr/bin/env python3
"""include/bpgpu.h -> the Rust `extern "C"` declarations of INTEGRATION.md section 1.
    python tools/gen_rust_extern.py            every prototype of the header as a Rust declaration
    python tools/gen_rust_extern.py --missing  only the ones INTEGRATION.md does not carry yet
(tests/test_abi_and_host.py checks that none is missing.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {"int": "c_int", "size_t": "usize", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "char": "c_char", "void": "c_void",
         "bpgpu_ctx": "bpgpu_ctx", "bpgpu_pool": "bpgpu_pool", "bpgpu_ticket": "bpgpu_ticket"}


def prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    for m in re.finditer(r"^\s*((?:const\s+)?[a-z_0-9]+\s*\**)\s*(bpgpu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        yield m.group(2), m.group(1).strip(), " ".join(m.group(3).split())


def rust_type(ctype):
    ctype = ctype.strip()
    arr = re.search(r"\[\d*\]$", ctype)
    if arr:
        ctype = ctype[:arr.start()] + "*"
    stars = ctype.count("*")
    words = ctype.replace("*", " ").split()
    const = "const" in words
    base = [w for w in words if w != "const"][0]
    t = TYPES[base]
    for i in range(stars):
        # `const T *const *p`: pointer to const pointers to const T
        t = ("*const " if const else "*mut ") + t
    return t


def rust_decl(name, ret, args):
    params = []
    if args and args != "void":
        for a in args.split(","):
            a = a.strip()
            m = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)(\[\d*\])?$", a)
            ctype, pname, arr = m.group(1), m.group(2), m.group(3) or ""
            low = {"type": "type_", "in": "in_", "ref": "ref_", "box": "box_"}.get(pname, pname).lower()
            if low in [q[0] for q in params]:      # `d_B` (a point) beside `d_b` (a vector): the upper-case one is the point
                low = low + ("_point" if pname != pname.lower() else "_vec")
            params.append((low, rust_type(ctype + arr)))
        params = ["%s: %s" % q for q in params]
    r = "" if ret == "void" else " -> " + rust_type(ret)
    return "    pub fn %s(%s)%s;" % (name, ", ".join(params), r)


def main():
    hdr = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    have = set(re.findall(r"fn (bpgpu_[a-z0-9_]+)", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    for name, ret, args in prototypes(hdr):
        if "--missing" in sys.argv and name in have:
            continue
        print(rust_decl(name, ret, args))


if __name__ == "__main__":
    main()
