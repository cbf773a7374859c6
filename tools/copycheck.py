#!/usr/bin/env python3
"""Mechanical copy check (container only: reads /root/reference): for every source file of this repository, the number of
its lines that are verbatim lines of a reference source file after whitespace normalisation (lines of >= 25 characters).
  python tools/copycheck.py [threshold]      lists files with at least `threshold` such lines (default 5)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
def norm(line):
    return re.sub(r"\s+", "", line)
ref_lines = set()
for d, _, fs in os.walk(REF):
    for f in fs:
        if f.endswith((".cpp", ".h", ".hpp", ".glsl", ".py", ".gd")):
            for l in open(os.path.join(d, f), errors="replace"):
                n = norm(l)
                if len(n) >= 25:
                    ref_lines.add(n)
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 5
files = subprocess.check_output(["git", "ls-files"], cwd=ROOT, text=True).split()
files += [f for f in subprocess.check_output(["git", "ls-files", "--others", "--exclude-standard"], cwd=ROOT, text=True).split()]
rows = []
for f in files:
    if not f.endswith((".cpp", ".h", ".hpp", ".hip", ".c", ".py", ".glsl")):
        continue
    lines = [norm(l) for l in open(os.path.join(ROOT, f), errors="replace")]
    lines = [l for l in lines if len(l) >= 25]
    hits = sum(1 for l in lines if l in ref_lines)
    if hits >= thr:
        rows.append((hits, len(lines), f))
for hits, n, f in sorted(rows, reverse=True):
    print("%4d of %4d substantial lines verbatim in the reference: %s" % (hits, n, f))
if not rows:
    print("no file has %d or more verbatim reference lines" % thr)
