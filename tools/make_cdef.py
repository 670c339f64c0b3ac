#!/usr/bin/env python3
"""include/ovqe_sv.h -> include/ovqe_sv.cdef.h: the same declarations without any preprocessor line, for
`cffi.FFI().cdef(open("include/ovqe_sv.cdef.h").read())` (cffi's cdef does not take #include, #ifdef or parenthesised
#define values).  Comments are dropped, the OVQE_* constants become one anonymous enum, <stdint.h> types are left to
cffi (it knows them).  Run after every change of the header; tests/test_abi.py checks that the committed file is current."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def generate(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    consts, body, skip = [], [], 0
    for line in text.splitlines():
        st = line.strip()
        if st.startswith("#ifdef __cplusplus"):
            skip += 1
            continue
        if skip and st.startswith("#endif"):
            skip -= 1
            continue
        if skip:
            continue
        m = re.match(r"#\s*define\s+(OVQE_[A-Z0-9_]+)\s+\(?(-?\d+)\)?\s*$", st)
        if m and m.group(1) != "OVQE_SV_H":
            consts.append((m.group(1), int(m.group(2))))
            continue
        if st.startswith("#"):
            continue
        body.append(line.rstrip())
    decls = re.sub(r"\n\s*\n+", "\n", "\n".join(body)).strip()
    out = ["/* generated from include/ovqe_sv.h by tools/make_cdef.py: do not edit.  For cffi: ffi.cdef(this file). */",
           "enum {"]
    out += [f"    {name} = {value}," for name, value in consts]
    out += ["};", decls, ""]
    return "\n".join(out)


def main():
    src = os.path.join(ROOT, "include", "ovqe_sv.h")
    dst = os.path.join(ROOT, "include", "ovqe_sv.cdef.h")
    text = generate(open(src).read())
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(dst) and open(dst).read() == text else 1)
    open(dst, "w").write(text)
    print("wrote", dst, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
