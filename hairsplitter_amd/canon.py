"""Order-insensitive canonical form of .col / .gro / .vcf files.

The reference writes contig blocks in hash-map / thread-completion order
(call_variants.cpp:1179,1274,1354; separate_reads.cpp:1754-1757), so files are compared per CONTIG
block keyed by contig name, blank lines dropped.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List


def split_blocks(path: str) -> Dict[str, List[str]]:
    blocks: Dict[str, List[str]] = {}
    cur = None
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("CONTIG"):
                name = line.split("\t")[1]
                cur = blocks.setdefault(name, [])
            if cur is None:
                cur = blocks.setdefault("", [])
            cur.append(line)
    return blocks


def vcf_blocks(path: str) -> Dict[str, List[str]]:
    blocks: Dict[str, List[str]] = {}
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("#"):
                continue
            blocks.setdefault(line.split("\t")[0], []).append(line)
    return blocks


def digest(blocks: Dict[str, List[str]]) -> str:
    h = hashlib.sha256()
    for name in sorted(blocks):
        h.update(name.encode() + b"\0")
        for l in blocks[name]:
            h.update(l.encode() + b"\n")
    return h.hexdigest()


def block_digests(blocks: Dict[str, List[str]]) -> Dict[str, str]:
    return {k: hashlib.sha256("\n".join(v).encode()).hexdigest() for k, v in blocks.items()}


def diff_blocks(a: Dict[str, List[str]], b: Dict[str, List[str]], limit: int = 5) -> List[str]:
    out = []
    for k in sorted(set(a) | set(b)):
        if k not in a or k not in b:
            out.append(f"contig {k!r} only in {'first' if k in a else 'second'}")
            continue
        la, lb = a[k], b[k]
        if la == lb:
            continue
        if len(la) != len(lb):
            out.append(f"contig {k!r}: {len(la)} vs {len(lb)} lines")
        for i, (x, y) in enumerate(zip(la, lb)):
            if x != y:
                out.append(f"contig {k!r} line {i}: {x[:160]!r} != {y[:160]!r}")
                if len(out) >= limit:
                    return out
                break
    return out
