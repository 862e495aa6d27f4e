#!/usr/bin/env python3
"""sha256 over the library's sources (csrc/*.hip, csrc/*.h, include/*.h, in name order): names the BINARY a profile pass was
taken on without needing git (the GPU box gets no .git).  scripts/pmc_traffic.py records it; bench.py compares it with the tree
it runs from and flags `traffic_stale` when they differ."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16(root=ROOT):
    h = hashlib.sha256()
    dirs = [os.path.join(root, "arrowspace-rs_amd", "csrc"), os.path.join(root, "include")]
    for d in dirs:
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h")):
                h.update(name.encode())
                with open(os.path.join(d, name), "rb") as f:
                    h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
