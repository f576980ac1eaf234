#!/usr/bin/env python3
"""Compare the gfx950 ISA of every kernel two device-assembly files have in common (hipcc -S --cuda-device-only).
Used to check that a change to a templated kernel leaves the instantiations it does not concern bit-identical.
    python tools/isa_diff.py old.s new.s [substring-of-demangled-name]"""
import re
import subprocess
import sys


def bodies(path):
    out, name, buf = {}, None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, buf = m.group(1), []
            continue
        if name is not None:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                out[name] = buf
                name = None
                continue
            t = line.split(";")[0].rstrip()
            if t and not t.lstrip().startswith("."):
                buf.append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return out


def main():
    a, b = bodies(sys.argv[1]), bodies(sys.argv[2])
    pat = sys.argv[3] if len(sys.argv) > 3 else ""
    names = sorted(set(a) & set(b))
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    same = diff = 0
    for n, d in zip(names, dem):
        if pat and pat not in d:
            continue
        if a[n] == b[n]:
            same += 1
        else:
            diff += 1
            print(f"DIFF {len(a[n])} -> {len(b[n])} instructions: {d[:200]}")
    print(f"{same} kernels identical, {diff} differ; only in old: {len(set(a) - set(b))}, only in new: {len(set(b) - set(a))}")


if __name__ == "__main__":
    main()
