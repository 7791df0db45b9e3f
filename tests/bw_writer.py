"""Minimal BigWig WRITER for tests (synthetic files for the section decoder): one-level
chromosome B+ tree, one-level-per-256 R-tree, sections of type 1/2/3, optional zlib."""
import struct
import zlib


def _section(cid, recs, typ):
    cstart, cend = recs[0][0], recs[-1][1]
    if typ == 1:
        body = b"".join(struct.pack("<IIf", a, b, v) for a, b, v in recs)
        step = span = 0
    elif typ == 2:
        span = recs[0][1] - recs[0][0]
        body = b"".join(struct.pack("<If", a, v) for a, _, v in recs)
        step = 0
    else:
        span = recs[0][1] - recs[0][0]
        step = recs[1][0] - recs[0][0] if len(recs) > 1 else span
        body = b"".join(struct.pack("<f", v) for _, _, v in recs)
    return struct.pack("<IIIIIBBH", cid, cstart, cend, step, span, typ, 0, len(recs)) + body


def _pick_type(recs, mix):
    if not mix:
        return 1
    spans = {b - a for a, b, _ in recs}
    if len(spans) == 1:
        steps = {recs[k + 1][0] - recs[k][0] for k in range(len(recs) - 1)}
        if len(steps) <= 1:
            return 3
        return 2
    return 1


def write_bigwig(path, chroms, data, items_per_block=512, compress=True, mix_types=False, pad=0):
    """pad: bytes of padding behind every compressed section that the index leaf's size INCLUDES (zlib readers stop at the
    end of the stream; a device decoder must find the Adler-32 trailer where the stream ends, not where the leaf ends)."""
    names = sorted(chroms)
    ids = {c: i for i, c in enumerate(names)}
    key = max(len(c) for c in names)
    sections = []       # (cid, start, end, bytes)
    for c in names:
        recs = data.get(c, [])
        k = 0
        while k < len(recs):
            chunk = recs[k:k + items_per_block]
            if mix_types:
                # split so that runs of equal span form type 2/3 candidates
                j = 1
                while j < len(chunk) and (chunk[j][1] - chunk[j][0]) == (chunk[0][1] - chunk[0][0]):
                    j += 1
                if j >= 3:
                    chunk = chunk[:j]
            raw = _section(ids[c], chunk, _pick_type(chunk, mix_types))
            sections.append((ids[c], chunk[0][0], chunk[-1][1], (zlib.compress(raw) + b"\xa5" * pad) if compress else raw, len(raw)))
            k += len(chunk)
    ubuf = max([s[4] for s in sections] + [1]) if compress else 0
    header_size = 64
    chrom_tree_off = header_size
    tree = struct.pack("<IIIIQQ", 0x78CA8C91, len(names), key, 8, len(names), 0)
    tree += struct.pack("<BBH", 1, 0, len(names))
    for c in names:
        tree += c.encode().ljust(key, b"\0") + struct.pack("<II", ids[c], chroms[c])
    data_off = chrom_tree_off + len(tree)
    blob = struct.pack("<Q", len(sections))
    offs = []
    pos = data_off + 8
    for s in sections:
        offs.append(pos)
        blob += s[3]
        pos += len(s[3])
    index_off = pos
    # R-tree: leaves of <= 256 items under one root (two levels when needed)
    leaves = [list(range(i, min(i + 256, len(sections)))) for i in range(0, len(sections), 256)] or [[]]
    hdr = struct.pack("<IIQIIIIQII", 0x2468ACE0, 256, len(sections),
                      sections[0][0] if sections else 0, sections[0][1] if sections else 0,
                      sections[-1][0] if sections else 0, sections[-1][2] if sections else 0, index_off, 1, 0)
    if len(leaves) == 1:
        node = struct.pack("<BBH", 1, 0, len(leaves[0]))
        for i in leaves[0]:
            s = sections[i]
            node += struct.pack("<IIIIQQ", s[0], s[1], s[0], s[2], offs[i], len(s[3]))
        index = hdr + node
    else:
        root_size = 4 + 24 * len(leaves)
        leaf_blobs, leaf_offs = [], []
        p = index_off + len(hdr) + root_size
        for lf in leaves:
            nb = struct.pack("<BBH", 1, 0, len(lf))
            for i in lf:
                s = sections[i]
                nb += struct.pack("<IIIIQQ", s[0], s[1], s[0], s[2], offs[i], len(s[3]))
            leaf_offs.append(p)
            leaf_blobs.append(nb)
            p += len(nb)
        root = struct.pack("<BBH", 0, 0, len(leaves))
        for lf, lo in zip(leaves, leaf_offs):
            a, b = sections[lf[0]], sections[lf[-1]]
            root += struct.pack("<IIIIQ", a[0], a[1], b[0], b[2], lo)
        index = hdr + root + b"".join(leaf_blobs)
    header = struct.pack("<IHHQQQHHQQIQ", 0x888FFC26, 4, 0, chrom_tree_off, data_off, index_off, 0, 0, 0, 0, ubuf, 0)
    with open(path, "wb") as fh:
        fh.write(header + tree + blob + index)
