#!/usr/bin/env python3
"""tools/fold_key.py — fold a commitment key for callers outside Python (C: b3w_commit_key_create_folded, Node: wc.setCommitKey(...,
folded)).  Needs no GPU: the slot widths come from the circuit's layout file and a table of atom widths measured once.

    python tools/fold_key.py <circuit> <curve> <generators.bin> <out prefix> [--first-slot N] [--r1cs file] [--widths file]

generators.bin: 64 bytes per committed slot (x, y little-endian).  Writes <out>.gens (folded generators, same format) and <out>.mask
(one byte per committed slot: 0 kept, 1 folded away, 0x80 | i only bit i of the word).  --widths: a file of uint16 little-endian
bit widths per witness slot as b3w_slot_widths returns them (default: ask the library — that needs a GPU for the context)."""
import argparse, importlib, os, sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("circuit"); ap.add_argument("curve"); ap.add_argument("generators"); ap.add_argument("out")
    ap.add_argument("--first-slot", type=int, default=0)
    ap.add_argument("--r1cs", default=None)
    ap.add_argument("--widths", default=None)
    a = ap.parse_args()
    m = importlib.import_module("hot-proofs-blake3-circom_amd")
    F = importlib.import_module("hot-proofs-blake3-circom_amd.fold")
    if a.widths:
        widths = np.fromfile(a.widths, dtype="<u2")
    else:
        ctx = m.Context(a.circuit, 0)
        widths = ctx.slot_widths()
        ctx.close()
    image = m.read_r1cs_image(a.r1cs or m.BUILTIN_R1CS[a.circuit])
    gens = open(a.generators, "rb").read()
    buf, mask, stats = F.fold_generators(image, widths, a.first_slot, gens, a.curve)
    open(a.out + ".gens", "wb").write(buf)
    open(a.out + ".mask", "wb").write(bytes(mask))
    print(stats)


if __name__ == "__main__":
    main()
