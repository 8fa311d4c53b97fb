"""Element table and chemical-shift standardisation constants.

The reference obtains these from the external ``nmrdata`` package (``load_embeddings()``,
``load_standards()``; nmrgnn/model.py:39,47, nmrgnn/library.py:34), which is not in the tree.
What the reference's bundled SavedModel pins (tests/golden/savedmodel_constants.json): a one-hot
width of 10 and (avg, std) = C (126.0, 10.603463), N (118.955, 50.941216), H (5.63, 6.040644) at
indices 2, 3, 4; every other index has std = avg = 0 (so those elements always predict 0).
The NAMES of the other seven slots are this package's convention (parity unpinned)."""
from __future__ import annotations

ELEMENTS = ['X', 'Z', 'C', 'N', 'H', 'O', 'S', 'P', 'F', 'Cl']
NUM_ELEM = len(ELEMENTS)

_STANDARDS = {
    2: ('C', 126.0, 10.603463172912598),
    3: ('N', 118.95500183105469, 50.94121551513672),
    4: ('H', 5.630000114440918, 6.04064416885376),
}


def load_embeddings():
    """{'atom': {symbol: index}} — the part of nmrdata.load_embeddings() the hot path uses."""
    return {'atom': {s: i for i, s in enumerate(ELEMENTS)}}


def load_standards():
    """{element index: (name, avg, std)} as nmrdata.load_standards() returns."""
    out = {}
    for i, s in enumerate(ELEMENTS):
        out[i] = _STANDARDS.get(i, (s, 0.0, 0.0))
    return out
