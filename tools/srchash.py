#!/usr/bin/env python3
"""Identity of the kernel sources a hardware-counter measurement was taken on.

``source_digest()`` = sha1 over the git blob hashes of every file that is compiled into libmrirt.so
(csrc/*.hip, csrc/*.h, include/mrirt.h).  profiles/traffic.json stores the digest next to each PMC entry;
bench.py emits ``roofline.traffic`` only when the tree it runs from has the same digest — counters measured on
other sources are stale by definition (the JSON went stale silently four commits after it was made).
"""
from __future__ import annotations

import hashlib
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent


def git_blob_sha1(data: bytes) -> str:
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_files():
    csrc = ROOT / "mri-raytracer_amd" / "csrc"
    return sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.h")) + [ROOT / "include" / "mrirt.h"])


def source_blobs() -> dict:
    return {str(p.relative_to(ROOT)): git_blob_sha1(p.read_bytes()) for p in source_files()}


def source_digest() -> str:
    h = hashlib.sha1()
    for name, blob in sorted(source_blobs().items()):
        h.update(f"{name}:{blob}\n".encode())
    return h.hexdigest()


if __name__ == "__main__":
    for k, v in source_blobs().items():
        print(v, k)
    print(source_digest(), "digest")
