#!/usr/bin/env python3
"""Digest of every kernel in gfx950 assembly files (`make asm`): name -> (instructions, sha1 of the normalised body).

Used to check that a change of the source layout (moving kernels between translation units, splitting headers) left the
generated code of every kernel untouched:   python tools/asm_digest.py kokoro-align_amd/csrc/*.s > after.txt; diff before.txt after.txt
Normalisation: comments and directives dropped, local labels renumbered in order of first appearance."""
import hashlib
import re
import sys


def digest(paths):
    out = {}
    for path in paths:
        txt = open(path, errors="replace").read()
        for k in re.split(r"\n(?=_Z\w+:)", txt):
            m = re.match(r"(_Z\w+):", k)
            if not m or "s_endpgm" not in k:
                continue
            body = k.split("\n")[1:]
            labels, lines = {}, []
            for ln in body:
                ln = ln.split(";")[0].rstrip()
                if not ln.strip():
                    continue
                s = ln.strip()
                if s.startswith(".") and not re.match(r"\.L\w+:", s):
                    if s.startswith((".section", ".rodata", ".amdhsa_kernel")):
                        break
                    continue
                lines.append(s)
            text = "\n".join(lines)
            for lab in re.findall(r"\.L\w+", text):
                labels.setdefault(lab, f".L{len(labels)}")
            text = re.sub(r"\.L\w+", lambda mm: labels[mm.group(0)], text)
            n = sum(1 for ln in lines if not ln.endswith(":"))
            out[m.group(1)] = (n, hashlib.sha1(text.encode()).hexdigest()[:12])
    return out


if __name__ == "__main__":
    for name, (n, h) in sorted(digest(sys.argv[1:]).items()):
        print(f"{name} {n} {h}")
