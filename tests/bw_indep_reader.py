"""A deliberately INDEPENDENT BigWig reader (test infrastructure): Python `struct` + `zlib`, written from the
published format description (Kent et al. 2010, "BigWig and BigBed", supplementary tables: common header, zoom
headers, chromosome B+ tree, R-tree index, section header + items) -- it shares no code and no traversal order with
csrc/wt_bigwig.cpp, tests/bw_writer.py or wiggletools_amd/bwwrite.py, so decoder and writers stop vouching for each
other.  Queries walk the R-tree with OVERLAP PRUNING per chromosome (what libBigWig's
bwOverlappingIntervals does for the reference, src/bigWiggleReader.c:58), items come back 0-based half-open, sorted by
start, as (start, end, value) numpy arrays."""
import struct
import zlib

import numpy as np

BIGWIG_MAGIC = 0x888FFC26
CHROM_TREE_MAGIC = 0x78CA8C91
RTREE_MAGIC = 0x2468ACE0


class IndependentBigWig:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        (magic, self.version, self.n_zoom, chrom_tree, self.data_off, self.index_off, self.field_count,
         self.defined_fields, self.autosql, self.summary_off, self.uncompress_buf, _res) = struct.unpack_from("<IHHQQQHHQQIQ", b, 0)
        if magic != BIGWIG_MAGIC:
            raise ValueError("not a little-endian BigWig file")
        # zoom headers follow the common header: reductionLevel, reserved, dataOffset, indexOffset (24 bytes each)
        self.zooms = [struct.unpack_from("<IIQQ", b, 64 + 24 * k) for k in range(self.n_zoom)]
        # chromosome B+ tree
        magic, block_size, self.key_size, val_size, n_items, _r = struct.unpack_from("<IIIIQQ", b, chrom_tree)
        if magic != CHROM_TREE_MAGIC or val_size != 8:
            raise ValueError("bad chromosome tree")
        self.chroms = {}        # name -> (id, length)
        self._bpt(chrom_tree + 32)
        if len(self.chroms) != n_items:
            raise ValueError("chromosome tree item count mismatch")
        (self.n_sections,) = struct.unpack_from("<Q", b, self.data_off)
        magic = struct.unpack_from("<I", b, self.index_off)[0]
        if magic != RTREE_MAGIC:
            raise ValueError("bad R-tree magic")
        self.rtree_root = self.index_off + 48

    def _bpt(self, off):
        is_leaf, _pad, count = struct.unpack_from("<BBH", self.b, off)
        off += 4
        for _ in range(count):
            key = self.b[off:off + self.key_size].split(b"\0", 1)[0].decode()
            if is_leaf:
                cid, size = struct.unpack_from("<II", self.b, off + self.key_size)
                self.chroms[key] = (cid, size)
            else:
                (child,) = struct.unpack_from("<Q", self.b, off + self.key_size)
                self._bpt(child)
            off += self.key_size + 8

    def _leaves(self, off, cid, lo, hi, out):
        """Index leaves overlapping chromosome `cid`, bases [lo, hi): (offset, size) in file order of the walk."""
        is_leaf, _pad, count = struct.unpack_from("<BBH", self.b, off)
        off += 4
        for _ in range(count):
            sc, sb, ec, eb = struct.unpack_from("<IIII", self.b, off)
            overlaps = (sc, sb) < (cid, hi) and (ec, eb) > (cid, lo)
            if is_leaf:
                d_off, d_size = struct.unpack_from("<QQ", self.b, off + 16)
                if overlaps:
                    out.append((d_off, d_size))
                off += 32
            else:
                (child,) = struct.unpack_from("<Q", self.b, off + 16)
                if overlaps:
                    self._leaves(child, cid, lo, hi, out)
                off += 24

    def section(self, d_off, d_size):
        raw = self.b[d_off:d_off + d_size]
        if self.uncompress_buf:
            raw = zlib.decompress(raw)
        cid, c_start, c_end, step, span, typ, _r, count = struct.unpack_from("<IIIIIBBH", raw, 0)
        body = memoryview(raw)[24:]
        if typ == 1:        # bedGraph: start, end, value
            a = np.frombuffer(body, dtype=np.dtype([("s", "<u4"), ("e", "<u4"), ("v", "<f4")]), count=count)
            s, e, v = a["s"].astype(np.int64), a["e"].astype(np.int64), a["v"].copy()
        elif typ == 2:      # variableStep: start, value; span from the header
            a = np.frombuffer(body, dtype=np.dtype([("s", "<u4"), ("v", "<f4")]), count=count)
            s = a["s"].astype(np.int64)
            e, v = s + span, a["v"].copy()
        elif typ == 3:      # fixedStep: value; start / step / span from the header
            v = np.frombuffer(body, dtype="<f4", count=count).copy()
            s = c_start + step * np.arange(count, dtype=np.int64)
            e = s + span
        else:
            raise ValueError("unknown section type %d" % typ)
        return cid, s, e, v

    def intervals(self, chrom, lo=0, hi=None):
        """(start0, end0, value) of the items overlapping [lo, hi) of `chrom`, sorted by start."""
        if chrom not in self.chroms:
            z = np.zeros(0, np.int64)
            return z, z.copy(), np.zeros(0, np.float32)
        cid, length = self.chroms[chrom]
        hi = length if hi is None else hi
        leaves = []
        self._leaves(self.rtree_root, cid, lo, hi, leaves)
        S, E, V = [], [], []
        for d_off, d_size in leaves:
            c, s, e, v = self.section(d_off, d_size)
            if c != cid:
                continue
            keep = (e > lo) & (s < hi)
            S.append(s[keep]); E.append(e[keep]); V.append(v[keep])
        if not S:
            z = np.zeros(0, np.int64)
            return z, z.copy(), np.zeros(0, np.float32)
        s, e, v = np.concatenate(S), np.concatenate(E), np.concatenate(V)
        order = np.argsort(s, kind="stable")
        return s[order], e[order], v[order]


def parse_wig(path):
    """The reference's text fixtures (.wig: fixedStep / variableStep) as {chrom: (start0, end0, value)}."""
    out = {}
    mode = chrom = None
    start = step = span = 1
    for line in open(path):
        line = line.strip()
        if not line or line.startswith(("track", "#", "browser")):
            continue
        if line.startswith(("fixedStep", "variableStep")):
            f = dict(kv.split("=") for kv in line.split()[1:])
            mode, chrom = line.split()[0], f["chrom"]
            span = int(f.get("span", 1))
            start, step = int(f.get("start", 1)), int(f.get("step", 1))
            out.setdefault(chrom, [])
            continue
        if mode == "fixedStep":
            out[chrom].append((start - 1, start - 1 + span, float(line)))
            start += step
        else:
            p, v = line.split()
            out[chrom].append((int(p) - 1, int(p) - 1 + span, float(v)))
    return {c: (np.array([r[0] for r in rs], np.int64), np.array([r[1] for r in rs], np.int64), np.array([r[2] for r in rs], np.float32))
            for c, rs in out.items()}
