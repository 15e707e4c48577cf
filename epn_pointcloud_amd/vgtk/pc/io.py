"""load_ply for the ASCII vertex-only PLY files the path reads (kernel point sets); the reference uses
`plyfile` (vgtk/vgtk/pc/io.py:6-10), absent from this image, and only parses rows -- no arithmetic."""
import numpy as np


def load_ply(file_name, with_faces=False, with_color=False, with_normal=False):
    if with_faces or with_color or with_normal:
        raise NotImplementedError("only vertex positions are needed on the hot path")
    with open(file_name, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii").splitlines()
    if not any(l.startswith("format ascii") for l in header):
        raise NotImplementedError("binary PLY: ship the table as .npy instead (see vgtk/data)")
    nv = int([l for l in header if l.startswith("element vertex")][0].split()[2])
    rows = raw[end:].decode("ascii").split("\n")
    return np.array([[float(t) for t in rows[i].split()[:3]] for i in range(nv)])
