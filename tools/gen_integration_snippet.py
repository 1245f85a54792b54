"""Regenerate the ctypes `LcrConfig` block of INTEGRATION.md section 2 from the binding that is actually used
(gym_lowcostrobot_amd/_capi.py:LcrConfig), so the snippet a reference maintainer would copy cannot drift from
`struct lcr_config` (include/lcr.h) again.   python tools/gen_integration_snippet.py [--check]
tests/test_abi.py::test_integration_snippet_matches_the_struct parses the block and holds its size to lcr_config_default's struct_size."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = "class LcrConfig(ctypes.Structure):", "\n\ncfg = LcrConfig()"


def block():
    from gym_lowcostrobot_amd._capi import LcrConfig

    names = {ctypes.c_uint32: "ctypes.c_uint32", ctypes.c_int32: "ctypes.c_int32", ctypes.c_int64: "ctypes.c_int64",
             ctypes.c_uint64: "ctypes.c_uint64", ctypes.c_double: "ctypes.c_double"}
    items = [f'("{n}", {names[t]})' for n, t in LcrConfig._fields_]
    lines, cur = [], "    _fields_ = ["
    for it in items:
        if len(cur) + len(it) + 2 > 118:
            lines.append(cur.rstrip())
            cur = "                "
        cur += it + ", "
    lines.append(cur.rstrip(", ") + "]")
    return BEGIN + "      # mirrors struct lcr_config in include/lcr.h (generated: tools/gen_integration_snippet.py)\n" + "\n".join(lines)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    s = open(path).read()
    a, b = s.index(BEGIN), s.index(END)
    new = s[:a] + block() + s[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == s else 1)
    open(path, "w").write(new)
    print("INTEGRATION.md section 2 regenerated" if new != s else "INTEGRATION.md section 2 already current")


if __name__ == "__main__":
    main()
