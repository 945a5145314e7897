"""Writers of the two small sdsl-framed side files `spumoni run` loads (tests only):
<ref>.doc (src/doc_array.cpp:184-201) and <ref>.pmlnulldb/.msnulldb
(src/emp_null_database.cpp:82-110).  int_vector<>: u64 size in bits, u8 width, 64-bit words."""
import struct

import numpy as np


def _int_vector(vals, width):
    vals = [int(v) for v in vals]
    bits = len(vals) * width
    words = [0] * ((bits + 63) // 64)
    for i, v in enumerate(vals):
        bit = i * width
        wi, sh = bit >> 6, bit & 63
        words[wi] |= (v << sh) & 0xFFFFFFFFFFFFFFFF
        if sh + width > 64:
            words[wi + 1] |= v >> (64 - sh)
    return struct.pack("<QB", bits, width) + b"".join(struct.pack("<Q", w) for w in words)


def _width(vals):
    mx = max([1] + [int(v) for v in vals])
    w = 1
    while (1 << w) <= mx:
        w += 1
    return w


def write_doc_array(path, doc_start, doc_end):
    w = max(_width(doc_start), _width(doc_end))
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(doc_start)))
        f.write(_int_vector(doc_start, w))
        f.write(_int_vector(doc_end, w))


def write_null_db(path, percentile_value, stats):
    with open(path, "wb") as f:
        f.write(struct.pack("<Qddd", len(stats), 0.0, float(np.mean(stats)), float(percentile_value)))
        f.write(_int_vector(stats, _width(stats)))


# The writer of <ref>.thrbv.spumoni / <ref>.thrbv.ms lives in the package (build_index.py --serialized uses it)
from spumoni_amd.sdsl_streams import write_thrbv  # noqa: E402,F401
