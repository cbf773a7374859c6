"""One hash over the library's sources (csrc/*.hip, *.h, *.cpp): profiles/current_*.json and profiles/isa_cost.json are stamped with it,
and bench.py drops figures that were profiled on other kernels than the ones it is running."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gdpathtracing_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
