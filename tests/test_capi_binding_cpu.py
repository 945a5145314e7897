"""CPU tier: the Python binding (spumoni_amd/capi.py) -- argument order, dtypes, buffer shapes of every host-buffer entry
point -- exercised without a GPU: a subprocess loads tests/fake_device (the C-ABI answered by the CPU oracle; test
infrastructure) through SPUMONI_GPU_LIB and holds what the binding returns against the oracle's own Python module.
(What the real library computes is the GPU tier's business; a binding that passed its arguments in the wrong order or
width would fail here.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import numpy as np
from spumoni_amd import capi, synth
from tests import cases
import oracle

assert "fake-device" in capi.version()
raw, text = cases.real_case(21, 5000, list(b"ACGT"), ndocs=3)
rng = np.random.default_rng(4)
seqs, offs = cases.reads_mixed(rng, text, list(b"ACGT"), 120, 150, [ord("N")])
orc = oracle.OracleIndex.from_raw(raw)
ix = capi.Index.from_raw(raw, 0)
assert (ix.n, ix.r) == (raw.n, raw.r)
assert np.array_equal(ix.text(), text)
# PML: lengths, documents, classes, both widths, classification alone
pml, pdocs = orc.pml(seqs, offs, want_docs=True)
f, a, b, s = oracle.classify(pml, offs, 50, 4)
for bits in (32, 16):
    got = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True, classify=(50, 4), bits=bits)
    assert got["lengths"].dtype == (np.uint16 if bits == 16 else np.uint32)
    assert np.array_equal(got["lengths"], pml) and np.array_equal(got["docs"], pdocs)
    assert np.array_equal(got["class"]["above"], a) and np.array_equal(got["class"]["below"], b) and np.array_equal(got["class"]["sum_max"], s)
only = ix.query_host(capi.SPX_MODE_PML, seqs, offs, want_lengths=False, classify=(50, 4))
assert "lengths" not in only and np.array_equal(only["class"]["above"], a)
# MS: pointers, lengths (text), documents
ms = orc.ms(seqs, offs, want_docs=True, text=text)
got = ix.query_host(capi.SPX_MODE_MS, seqs, offs, want_docs=True, classify=(60, 6))
assert np.array_equal(got["pointers"], ms["pointers"]) and np.array_equal(got["lengths"], ms["lengths"]) and np.array_equal(got["docs"], ms["docs"])
f2, a2, b2, s2 = oracle.classify(ms["lengths"], offs, 60, 6)
assert np.array_equal(got["class"]["above"], a2) and np.array_equal(got["class"]["sum_max"], s2)
# the vectors as text: gap + "v v v \n" per read, offsets of every record
gap = np.array([len("read_%d" % q) + 2 for q in range(offs.size - 1)], dtype=np.uint32)
tx = ix.query_text(capi.SPX_MODE_MS, seqs, offs, gap=gap, streams=1 | 2 | 4, classify=(60, 6))
for i, vals in enumerate((ms["lengths"], ms["pointers"], ms["docs"])):
    ls = tx["line_start"][i]
    body = tx["text"][i]
    assert int(ls[-1]) == len(body)
    for q in (0, 1, 57, offs.size - 2):
        line = "".join("%d " % v for v in vals[offs[q]:offs[q + 1]]) + "\n"
        assert body[int(ls[q]) + int(gap[q]): int(ls[q + 1])].decode() == line, (i, q)
assert np.array_equal(tx["class"]["above"], a2)
# digestion: the host forms
for kind, k, w in ((1, 4, 11), (2, 3, 5)):
    dseq, doff = ix.digest_host(kind, k, w, np.frombuffer(seqs.tobytes().upper(), dtype=np.uint8), offs)
    want = [oracle.digest(kind, k, w, np.frombuffer(seqs[offs[q]:offs[q + 1]].tobytes().upper(), dtype=np.uint8)) for q in range(offs.size - 1)]
    assert np.array_equal(np.diff(doff.astype(np.int64)), [len(x) for x in want])
    assert dseq.tobytes() == b"".join(bytes(x) for x in want)
up = np.frombuffer(seqs.tobytes().upper(), dtype=np.uint8)
dq = ix.digest_query_host(capi.SPX_MODE_PML, 2, 3, 5, up, offs, classify=(50, 4))
dseq, doff = ix.digest_host(2, 3, 5, up, offs)
assert np.array_equal(dq["offsets"], doff) and np.array_equal(dq["lengths"], orc.pml(dseq, doff))
# handles: tag, clone, options, errors
ix.set_source_tag("files-of-today")
cl = ix.clone(0)
assert cl.source_tag() == "files-of-today" and np.array_equal(cl.query_host(capi.SPX_MODE_PML, seqs, offs)["lengths"], pml)
ix.set_option("minimizer_charhash", 0x01020304)
r2, _ = cases.real_case(22, 800, list(b"ACGT"))
plain = capi.Index.from_raw(synth.RawIndex(heads=r2.heads, lens=r2.lens, thr=r2.thr, n=r2.n), 0)  # no samples, no documents
for call in (lambda: plain.query_host(capi.SPX_MODE_MS, seqs, offs), lambda: plain.query_host(capi.SPX_MODE_PML, seqs, offs, want_docs=True),
             lambda: ix.query_host(capi.SPX_MODE_PML, np.full(70000, 65, np.uint8), np.array([0, 70000]), bits=16)):
    try:
        call()
        raise SystemExit("an error was expected")
    except capi.SpxError:
        pass
print("BINDING OK")
'''


def test_python_binding_marshals_every_host_entry_point(fake_device, oracle_mod):
    env = dict(os.environ, SPUMONI_GPU_LIB=os.path.join(fake_device, "libspumoni_gpu.so"), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "BINDING OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
