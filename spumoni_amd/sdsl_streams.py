"""Writer of the serialised index files `spumoni run` loads -- <ref>.thrbv.spumoni / <ref>.thrbv.ms
(/root/reference/src/compute_ms_pml.cpp:192-213, 517-542: terminator position, F, the rle_string, [samples_last,] the
per-letter threshold vectors, [samples_start]) -- in the sdsl-lite / r-index stream layout restated in
spumoni_amd/csrc/host/index_files.cpp.  UNVERIFIED against an upstream-built file: select supports are written with the
right framing but placeholder contents (readers that rebuild them, like ours, do not look inside), so the files are for
our reader and for diffing structure with tools/pin_upstream.py, not byte-identical to `spumoni build`'s.  Plain Python
loops: meant for small and medium indexes (build_index.py --serialized)."""
import struct


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


def _bit_vector(bits):
    n = len(bits)
    words = [0] * ((n + 63) // 64)
    for i, b in enumerate(bits):
        if b:
            words[i >> 6] |= 1 << (i & 63)
    return struct.pack("<Q", n) + b"".join(struct.pack("<Q", w) for w in words)


def _select_mcl(arg_cnt):
    out = struct.pack("<Q", arg_cnt)
    if arg_cnt:
        sb = (arg_cnt + 4095) >> 12
        out += _int_vector([0] * sb, 8)  # superblock
        out += struct.pack("<Q", 0)  # empty mini_or_long: no long superblocks
        for _ in range(sb):
            out += _int_vector([0] * 64, 8)  # one miniblock vector per superblock
    return out


def _hi(x):
    return x.bit_length() - 1


def _sd_vector(ones, universe):
    n = len(ones)
    logm = _hi(universe) + 1 if universe else 0
    logn = _hi(n) + 1 if n else 0
    if logm == logn and logn > 0:
        logn -= 1
    wl = logm - logn
    low = [p & ((1 << wl) - 1) for p in ones]
    high = [0] * (n + (1 << logn))
    for i, p in enumerate(ones):
        high[(p >> wl) + i] = 1
    # width 0 is not a valid int_vector width: sdsl falls back to 64 bits per (all-zero) entry
    low_stream = _int_vector(low, wl) if wl > 0 else _int_vector([0] * n, 64)
    return (struct.pack("<QB", universe, wl) + low_stream + _bit_vector(high) + _select_mcl(n)
            + _select_mcl(len(high) - n))


def _sparse_sd(ones, universe):
    out = struct.pack("<QQ", universe, len(ones))
    if universe:
        out += _sd_vector(ones, universe)
    return out


def _wt_huff(seq):
    """A prefix-code wavelet tree over `seq` in wt_pc's stream framing (level order bit vector)."""
    import heapq

    size = len(seq)
    freq = {}
    for c in seq:
        freq[c] = freq.get(c, 0) + 1
    # Huffman tree: nodes as [weight, tiebreak, symbol or None, left, right]
    heap = [[w, c, c, None, None] for c, w in sorted(freq.items())]
    heapq.heapify(heap)
    tb = 256
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        heapq.heappush(heap, [a[0] + b[0], tb, None, a, b])
        tb += 1
    root = heap[0]
    # number nodes breadth-first, root = 0
    order, queue = [], [(root, 0xFFFF)]
    while queue:
        nd, par = queue.pop(0)
        nd.append(len(order))  # id
        nd.append(par)
        order.append(nd)
        if nd[2] is None:
            queue.append((nd[3], nd[-2]))
            queue.append((nd[4], nd[-2]))
    # bit vector: inner nodes in id order, each with the bits of its subsequence
    pos_of = {0: list(range(size))}
    bv, nodes = [], []
    for nd in order:
        nid = nd[-2]
        if nd[2] is not None:
            nodes.append((0, 0, nd[-1], 0xFFFF, 0xFFFF))
            continue
        syms_right = set()
        stack = [nd[4]]
        while stack:
            x = stack.pop()
            if x[2] is None:
                stack += [x[3], x[4]]
            else:
                syms_right.add(x[2])
        mine = pos_of[nid]
        bits = [1 if seq[p] in syms_right else 0 for p in mine]
        nodes.append((len(bv), sum(bv), nd[-1], nd[3][-2], nd[4][-2]))
        pos_of[nd[3][-2]] = [p for p, b in zip(mine, bits) if not b]
        pos_of[nd[4][-2]] = [p for p, b in zip(mine, bits) if b]
        bv += bits
    c_to_leaf = [0xFFFF] * 256
    for nd in order:
        if nd[2] is not None:
            c_to_leaf[nd[2]] = nd[-2]
    ones = sum(bv)
    out = struct.pack("<QQ", size, len(freq)) + _bit_vector(bv)
    out += struct.pack("<Q", 0)  # rank_support_v: (placeholder) empty int_vector<64>
    out += _select_mcl(ones) + _select_mcl(len(bv) - ones)
    out += struct.pack("<Q", len(nodes))
    for bv_pos, bv_rank, par, c0, c1 in nodes:
        out += struct.pack("<QQHHH", bv_pos, bv_rank, par, c0, c1)
    out += struct.pack("<256H", *c_to_leaf) + struct.pack("<256Q", *([0] * 256))
    return out


def write_thrbv(path, heads, lens, thr, ssa=None, esa=None):
    """<ref>.thrbv.spumoni (ssa is None) or <ref>.thrbv.ms, from raw per-run arrays."""
    heads = [max(int(h), 1) for h in heads]
    lens = [int(x) for x in lens]
    R, n = len(heads), sum(lens)
    F = [0] * 256
    for h, l in zip(heads, lens):
        F[h] += l
    acc, Fc = 0, []
    for c in range(256):
        Fc.append(acc)
        acc += F[c]
    # terminator_position: the BWT position of the terminator -- the start of the run whose head is the terminator (written
    # as 1 after the rewrite of compute_ms_pml.cpp:470-476; every text byte is >= 2)   (was a constant 0)
    term_pos, p0 = 0, 0
    for h, l in zip(heads, lens):
        if h <= 1:
            term_pos = p0
            break
        p0 += l
    out = struct.pack("<Q", term_pos) + struct.pack("<Q", 256) + struct.pack("<256Q", *Fc)
    # rle_string: n, R, B, runs (last position of every B-th run), runs_per_letter, run_heads
    B = 2
    out += struct.pack("<QQQ", n, R, B)
    ends, p = [], 0
    for l in lens:
        p += l
        ends.append(p - 1)
    out += _sparse_sd([e for i, e in enumerate(ends) if i % B == B - 1], n)
    per = {c: [] for c in range(256)}
    cnt = [0] * 256
    for h, l in zip(heads, lens):
        cnt[h] += l
        per[h].append(cnt[h] - 1)
    for c in range(256):
        out += _sparse_sd(per[c], cnt[c])
    out += _wt_huff(heads)
    logn = n.bit_length()
    if ssa is not None:
        out += _int_vector(esa, logn)  # samples_last
    # thr_bv: per letter, the non-zero thresholds in run order, universe n for letters that occur
    tl = {c: [] for c in range(256)}
    for h, t in zip(heads, thr):
        if int(t) > 0:
            tl[h].append(int(t))
    for c in range(256):
        out += _sparse_sd(tl[c], n if cnt[c] else 0)
    if ssa is not None:
        out += _int_vector(ssa, logn)  # samples_start
    with open(path, "wb") as f:
        f.write(out)
