"""Minimal BigWig WRITER over numpy arrays: bedGraph (type 1) sections of `items_per_block`
records, zlib level 1, chromosome B+ tree with one leaf, R-tree of <= 256 x 256 sections.  Exists so
that bench.py's file-to-result leg and the tests have real BigWig files to read (there is no network
for public tracks and the reference's fixtures are 10 bp long); the product reads BigWig, it never
writes it (the reference does not either: wigWriter.c emits text)."""
import ctypes as C
import struct
import zlib

import numpy as np


class FileSet:
    """N BigWig files written side by side by the library's native writer (csrc/wt_bwwrite.cpp): one chromosome of all
    tracks per call, the tracks dealt to worker threads -- what bench.py's whole-genome file leg writes its 100 x 24
    chromosome inputs with.  Chromosomes must be added in strcmp order of their names."""

    def __init__(self, paths, chroms, items_per_block=1024, level=1, threads=16):
        from . import _lib
        self.L = _lib.lib()
        self.L.wtamd_bw_writer_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        self.L.wtamd_bw_writers_add_chrom.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L.wtamd_bw_writer_close.argtypes = [C.c_void_p]
        self.L.wtamd_bw_writer_close.restype = C.c_int64
        self.names = sorted(chroms, key=lambda s: s.encode())
        arr = (C.c_char_p * len(self.names))(*[n.encode() for n in self.names])
        lens = np.array([int(chroms[n]) for n in self.names], np.uint32)
        self.threads = threads
        self.handles = (C.c_void_p * len(paths))()
        for i, p in enumerate(paths):
            h = C.c_void_p()
            if self.L.wtamd_bw_writer_open(str(p).encode(), len(self.names), arr, lens.ctypes.data, items_per_block, level, C.byref(h)) != 0:
                raise RuntimeError("wtamd_bw_writer_open(%s) failed" % p)
            self.handles[i] = h
        self.sections = None

    def add_chrom(self, name, seg_off, start, finish, value):
        """One chromosome of every file: track i = [seg_off[i], seg_off[i + 1]) of the (host) arrays -- int64 seg_off,
        int32 1-based start / exclusive finish, float32 value."""
        seg_off = np.ascontiguousarray(seg_off, np.int64)
        assert len(seg_off) == len(self.handles) + 1
        start = np.ascontiguousarray(start, np.int32); finish = np.ascontiguousarray(finish, np.int32)
        value = np.ascontiguousarray(value, np.float32)
        rc = self.L.wtamd_bw_writers_add_chrom(self.handles, len(self.handles), self.names.index(name), seg_off.ctypes.data,
                                               start.ctypes.data, finish.ctypes.data, value.ctypes.data, self.threads)
        if rc != 0:
            raise RuntimeError("wtamd_bw_writers_add_chrom(%s) failed: %d" % (name, rc))

    def add_chrom_ptr(self, name, seg_off, start_ptr, finish_ptr, value_ptr):
        """The same over raw host pointers (pinned staging of the generator's output)."""
        seg_off = np.ascontiguousarray(seg_off, np.int64)
        rc = self.L.wtamd_bw_writers_add_chrom(self.handles, len(self.handles), self.names.index(name), seg_off.ctypes.data,
                                               start_ptr, finish_ptr, value_ptr, self.threads)
        if rc != 0:
            raise RuntimeError("wtamd_bw_writers_add_chrom(%s) failed: %d" % (name, rc))

    def close(self):
        self.sections = [int(self.L.wtamd_bw_writer_close(h)) for h in self.handles]
        if min(self.sections) < 0:
            raise RuntimeError("wtamd_bw_writer_close failed")
        return self.sections

_REC = np.dtype([("s", "<u4"), ("e", "<u4"), ("v", "<f4")])


def write_arrays(path, chroms, data, items_per_block=1024, level=1):
    """chroms: {name: length}; data: {name: (start0, end0, value)} 0-based half-open numpy arrays."""
    names = sorted(chroms, key=lambda s: s.encode())
    key = max(len(c) for c in names)
    sections = []                               # (cid, start, end, blob, raw_len)
    for cid, c in enumerate(names):
        if c not in data:
            continue
        s0, e0, v = data[c]
        rec = np.empty(len(s0), _REC)
        rec["s"], rec["e"], rec["v"] = s0, e0, v
        for k in range(0, len(rec), items_per_block):
            chunk = rec[k:k + items_per_block]
            raw = struct.pack("<IIIIIBBH", cid, int(chunk["s"][0]), int(chunk["e"][-1]), 0, 0, 1, 0, len(chunk)) + chunk.tobytes()
            sections.append((cid, int(chunk["s"][0]), int(chunk["e"][-1]), zlib.compress(raw, level), len(raw)))
    if len(sections) > 65536:
        raise ValueError("too many sections for the two-level index: raise items_per_block")
    ubuf = max([s[4] for s in sections] + [1])
    tree = struct.pack("<IIIIQQ", 0x78CA8C91, len(names), key, 8, len(names), 0) + struct.pack("<BBH", 1, 0, len(names))
    for cid, c in enumerate(names):
        tree += c.encode().ljust(key, b"\0") + struct.pack("<II", cid, int(chroms[c]))
    data_off = 64 + len(tree)
    offs, pos = [], data_off + 8
    for s in sections:
        offs.append(pos)
        pos += len(s[3])
    index_off = pos
    first = sections[0] if sections else (0, 0, 0)
    last = sections[-1] if sections else (0, 0, 0)
    hdr = struct.pack("<IIQIIIIQII", 0x2468ACE0, 256, len(sections), first[0], first[1], last[0], last[2], index_off, 1, 0)
    leaves = [range(i, min(i + 256, len(sections))) for i in range(0, len(sections), 256)] or [range(0)]

    def leaf(lf):
        nb = struct.pack("<BBH", 1, 0, len(lf))
        for i in lf:
            s = sections[i]
            nb += struct.pack("<IIIIQQ", s[0], s[1], s[0], s[2], offs[i], len(s[3]))
        return nb

    if len(leaves) == 1:
        index = hdr + leaf(leaves[0])
    else:
        blobs = [leaf(lf) for lf in leaves]
        p = index_off + len(hdr) + 4 + 24 * len(leaves)
        root = struct.pack("<BBH", 0, 0, len(leaves))
        for lf, nb in zip(leaves, blobs):
            a, b = sections[lf[0]], sections[lf[-1]]
            root += struct.pack("<IIIIQ", a[0], a[1], b[0], b[2], p)
            p += len(nb)
        index = hdr + root + b"".join(blobs)
    header = struct.pack("<IHHQQQHHQQIQ", 0x888FFC26, 4, 0, 64, data_off, index_off, 0, 0, 0, 0, ubuf, 0)
    with open(path, "wb") as fh:
        fh.write(header + tree + struct.pack("<Q", len(sections)))
        for s in sections:
            fh.write(s[3])
        fh.write(index)
    return len(sections)
