"""Append-only ``.npy`` files for the chains' histories.

The reference rewrites every per-baseline file with everything sampled so far at each
``write_Niter`` flush (pspec.py:625-636 through utils.write_numpy_files, utils.py:272-312): O(Niter^2)
bytes over a run.  Here a history file is created once with a header that has room for the final
shape, each flush appends only the new rows and then rewrites the shape in the header IN PLACE.  At
every instant the file on disk is a valid ``.npy`` of shape ``(done, ...)`` -- exactly what the
reference leaves behind after a flush -- readable with ``numpy.load`` while the run goes on:

* the data of a flush are written (and, with ``fsync=True``, forced to disk) BEFORE the header
  changes; numpy reads ``prod(shape)`` items and ignores bytes beyond them, so a crash between the
  two leaves the previous valid prefix;
* the header rewrite is one ``pwrite`` of at most 128 bytes inside the first block of the file.

Format: version 1.0 (numpy.lib.format): magic ``\\x93NUMPY\\x01\\x00``, little-endian uint16 header
length, an ASCII dict literal padded with spaces and ended by a newline, total a multiple of 64.
"""
import os

import numpy as np

_MAGIC = b"\x93NUMPY\x01\x00"


def _header_bytes(dtype, shape, total_len=None):
    d = "{'descr': %r, 'fortran_order': False, 'shape': %r, }" % (np.lib.format.dtype_to_descr(np.dtype(dtype)),
                                                                  tuple(int(v) for v in shape))
    need = len(_MAGIC) + 2 + len(d) + 1
    if total_len is None:
        # room for a first dimension of up to 20 digits, as numpy's own writer leaves (GROWTH_AXIS_MAX_DIGITS)
        total_len = -(-(need + 20) // 64) * 64
    if need > total_len:
        return None
    hlen = total_len - len(_MAGIC) - 2
    return _MAGIC + int(hlen).to_bytes(2, "little") + (d + " " * (hlen - len(d) - 1) + "\n").encode("latin1")


def read_header(path, with_version=False):
    """(shape, dtype, data_offset) of a C-ordered ``.npy`` file (format 1.0 or 2.0)."""
    with open(path, "rb") as f:
        major, _ = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if major == 1
                                 else np.lib.format.read_array_header_2_0(f))
        if fortran and len(shape) > 1:
            raise ValueError(f"{path}: Fortran-ordered array cannot be appended to")
        if with_version:
            return tuple(shape), dtype, f.tell(), major
        return tuple(shape), dtype, f.tell()


class NpyAppender:
    """One history file: rows of ``row_shape`` / ``dtype`` appended flush by flush.

    ``start(keep_rows)``: create the file (``keep_rows`` None or 0 and no usable file), or continue an existing one cut
    back to its first ``keep_rows`` rows (a ``--resume``).  A file whose header cannot be rewritten in place (written by
    another tool without padding, or format 2.0) is converted once.  No file descriptor is kept between flushes: a
    run holds thousands of these."""

    def __init__(self, path, row_shape, dtype, fsync=False):
        self.path = os.fspath(path)
        self.row_shape = tuple(int(v) for v in row_shape)
        self.dtype = np.dtype(dtype)
        self.fsync = bool(fsync)
        self.rows = 0
        self.offset = None
        self.row_bytes = int(np.prod(self.row_shape, dtype=np.int64)) * self.dtype.itemsize

    def start(self, keep_rows=0):
        keep_rows = int(keep_rows or 0)
        if keep_rows > 0:
            shape, dtype, off, major = read_header(self.path, with_version=True)
            if tuple(shape[1:]) != self.row_shape or dtype != self.dtype:
                raise ValueError(f"{self.path}: holds {shape} {dtype}, expected (*, {self.row_shape}) {self.dtype}")
            if shape[0] < keep_rows:
                raise ValueError(f"{self.path}: holds {shape[0]} rows, {keep_rows} are needed")
            hdr = _header_bytes(self.dtype, (keep_rows,) + self.row_shape, total_len=off)
            if hdr is None or major != 1:
                data = np.load(self.path, mmap_mode="r")[:keep_rows]
                hdr = _header_bytes(self.dtype, (keep_rows,) + self.row_shape)
                tmp = self.path + ".conv.tmp"
                with open(tmp, "wb") as f:
                    f.write(hdr)
                    f.write(np.ascontiguousarray(data).tobytes())
                    f.flush()
                    if self.fsync:
                        os.fsync(f.fileno())
                del data
                os.replace(tmp, self.path)
                off = len(hdr)
            else:
                with open(self.path, "r+b") as f:      # shape first (a valid, shorter file), then cut the tail off
                    f.write(hdr)
                    f.flush()
                    if self.fsync:
                        os.fsync(f.fileno())
                    f.truncate(off + keep_rows * self.row_bytes)
            self.offset, self.rows = off, keep_rows
            return self
        hdr = _header_bytes(self.dtype, (0,) + self.row_shape)
        tmp = self.path + ".new.tmp"
        with open(tmp, "wb") as f:
            f.write(hdr)
        os.replace(tmp, self.path)
        self.offset, self.rows = len(hdr), 0
        return self

    def append(self, rows):
        """``rows``: array (k,) + row_shape, C-contiguous.  Data first, then the header's shape."""
        a = np.asarray(rows)
        if a.dtype != self.dtype or tuple(a.shape[1:]) != self.row_shape:
            raise ValueError(f"{self.path}: rows of {a.shape[1:]} {a.dtype} do not match (*, {self.row_shape}) {self.dtype}")
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)
        k = int(a.shape[0])
        if k == 0:
            return
        hdr = _header_bytes(self.dtype, (self.rows + k,) + self.row_shape, total_len=self.offset)
        assert hdr is not None, "header space exhausted"
        fd = os.open(self.path, os.O_RDWR)
        try:
            mv, pos = memoryview(a).cast("B"), self.offset + self.rows * self.row_bytes
            while len(mv):
                w = os.pwrite(fd, mv[:1 << 30], pos)
                mv, pos = mv[w:], pos + w
            if self.fsync:
                os.fsync(fd)
            os.pwrite(fd, hdr, 0)
            if self.fsync:
                os.fsync(fd)
        finally:
            os.close(fd)
        self.rows += k
