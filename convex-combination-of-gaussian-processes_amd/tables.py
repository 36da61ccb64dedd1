"""Reader / writer for the reference's text tables.

The bundled files come in two dialects, both CRLF-terminated:
  * `write.table` output: a quoted header line and a quoted row-name first column
    (Qian sets, hyperpars.matrix.txt, Ground-Vibrations sets; read by
    read.table(..., header = T) at HX:749-750, HX:765, GV:710-711, ADV:948);
  * bare tab-separated numbers, no header, possibly no trailing newline
    (maximin designs; read.table("maximin 14 pts.txt") at ADV:935).
"""
from __future__ import annotations

import numpy as np


def _tokens(line: str):
    return line.replace("\t", " ").split()


def read_table(path):
    """Return (column_names or None, float64 array [rows, cols]).  A first line made only
    of quoted fields is a header; when data rows have one more field than the header
    the first field of each row is a row name and is dropped (R's convention)."""
    with open(path, "r", newline="") as fh:
        lines = [ln.strip("\r\n") for ln in fh.read().split("\n")]
    lines = [ln for ln in lines if ln.strip()]
    names = None
    first = _tokens(lines[0])
    if all(tok.startswith('"') and tok.endswith('"') for tok in first):
        names = [tok.strip('"') for tok in first]
        lines = lines[1:]
    rows = []
    for ln in lines:
        toks = _tokens(ln)
        if names is not None and len(toks) == len(names) + 1:
            toks = toks[1:]
        elif toks and toks[0].startswith('"'):
            toks = toks[1:]
        rows.append([float(t) for t in toks])
    return names, np.asarray(rows, dtype=np.float64)


def write_table(path, array, names, row_names=None):
    """Same dialect as write.table(data.frame) (GV:760-761): quoted header, quoted
    row names, space separated, 15 significant digits."""
    array = np.asarray(array, dtype=np.float64)
    with open(path, "w", newline="") as fh:
        fh.write(" ".join('"%s"' % n for n in names) + "\r\n")
        for i, row in enumerate(array):
            rn = row_names[i] if row_names is not None else str(i + 1)
            fh.write('"%s" ' % rn + " ".join(repr(float(v)) if v != int(v) else str(int(v)) for v in row) + "\r\n")
