"""Minimal read-only HDF5 reader (pure Python + numpy), enough for UVH5 files.

The reference loads visibilities through pyuvdata/h5py (reference utils.py:105-199,
run-hydra-pspec.py:305-322); neither is a dependency of this package.  This module
reads the subset of the HDF5 file format that h5py's default writer produces (and
a little more):

* superblock versions 0-3; object headers versions 1 and 2;
* groups stored as symbol tables (B-tree v1 + local heap) or as compact link messages;
* datasets with compact, contiguous or chunked (B-tree v1 index) layout, filters
  deflate, shuffle, fletcher32 and LZF (h5py's filter 32000), honouring per-chunk
  filter masks;
* datatypes: integers, IEEE floats, enums (as their base integer), fixed-length strings
  and compounds of numeric members (``{r, i}`` compounds are returned as complex).

Not supported (raises ``NotImplementedError``): dense link storage (fractal heaps),
version-4 chunk indexes (``libver='latest'`` files), variable-length data, references.

Written from the public "HDF5 File Format Specification" (versions 1.1-3.0).
"""
import mmap
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


_LOW_LEVEL = (zlib.error, ValueError, OverflowError, IndexError, struct.error, RecursionError, MemoryError,
              UnicodeDecodeError, TypeError)


def _guard(fn):
    """Malformed files must fail with H5Error, whatever the decoder tripped over."""
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except _LOW_LEVEL as e:
            raise H5Error(f"corrupt or unsupported HDF5 structure ({type(e).__name__}: {e})") from e
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


def lzf_decompress(src, out_len):
    """LibLZF decompression (the algorithm behind HDF5 filter 32000)."""
    src = memoryview(src)
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:                       # literal run of ctrl + 1 bytes
            run = ctrl + 1
            if op + run > out_len or ip + run > n:
                raise H5Error("lzf: corrupt literal run")
            out[op:op + run] = src[ip:ip + run]
            ip += run
            op += run
        else:                               # back reference
            ln = ctrl >> 5
            ref = op - ((ctrl & 0x1F) << 8) - 1
            if ln == 7:
                ln += src[ip]
                ip += 1
            ref -= src[ip]
            ip += 1
            ln += 2
            if ref < 0 or op + ln > out_len:
                raise H5Error("lzf: corrupt back reference")
            if ref + ln <= op:
                out[op:op + ln] = out[ref:ref + ln]
            else:                           # overlapping copy: byte by byte semantics
                for k in range(ln):
                    out[op + k] = out[ref + k]
            op += ln
    if op != out_len:
        raise H5Error(f"lzf: produced {op} bytes, expected {out_len}")
    return bytes(out)


def _unshuffle(buf, itemsize):
    a = np.frombuffer(buf, dtype=np.uint8)
    n = len(a) // itemsize
    body = a[:n * itemsize].reshape(itemsize, n).T.reshape(-1)
    return body.tobytes() + a[n * itemsize:].tobytes()


class _Reader:
    def __init__(self, buf, so=8, sl=8):
        self.buf, self.so, self.sl = buf, so, sl

    def u(self, off, size):
        return int.from_bytes(self.buf[off:off + size], "little")

    def offs(self, off):
        return self.u(off, self.so)

    def lens(self, off):
        return self.u(off, self.sl)

    def cstr(self, off):
        end = self.buf.find(b"\x00", off)
        return bytes(self.buf[off:end]).decode("utf-8", "replace")


# ------------------------------------------------------------------ datatypes
def _parse_datatype(b, off):
    """-> (numpy dtype or descriptor, bytes consumed).  Compounds of two equal floats named
    r/i (or real/imag) become complex dtypes; enums become their base integer."""
    cv = b[off]
    cls, ver = cv & 0x0F, cv >> 4
    bits = b[off + 1] | (b[off + 2] << 8) | (b[off + 3] << 16)
    size = struct.unpack_from("<I", b, off + 4)[0]
    p = off + 8
    if cls == 0:                                   # fixed point
        order = ">" if bits & 1 else "<"
        signed = bool(bits & 8)
        return np.dtype(f"{order}{'i' if signed else 'u'}{size}"), p + 4 - off
    if cls == 1:                                   # floating point
        order = ">" if bits & 1 else "<"
        return np.dtype(f"{order}f{size}"), p + 12 - off
    if cls == 3:                                   # fixed-length string
        return np.dtype(f"S{size}"), p - off
    if cls == 4:                                   # bit field
        return np.dtype(f"<u{size}"), p + 4 - off
    if cls == 8:                                   # enumeration: base type, names, values
        nmem = bits & 0xFFFF
        base, used = _parse_datatype(b, p)
        p += used
        for _ in range(nmem):
            end = b.find(b"\x00", p)
            ln = end - p + 1
            p += ln if ver >= 3 else (ln + 7) // 8 * 8
        p += nmem * base.itemsize
        return base, p - off
    if cls == 6:                                   # compound
        nmem = bits & 0xFFFF
        names, offsets, types = [], [], []
        for _ in range(nmem):
            end = b.find(b"\x00", p)
            names.append(bytes(b[p:end]).decode())
            ln = end - p + 1
            p += ln if ver >= 3 else (ln + 7) // 8 * 8
            if ver >= 3:
                nb = 1
                while (1 << (8 * nb)) <= size and nb < 8:
                    nb += 1
                offsets.append(int.from_bytes(b[p:p + nb], "little"))
                p += nb
            else:
                offsets.append(struct.unpack_from("<I", b, p)[0])
                p += 4
                if ver == 1:
                    p += 1 + 3 + 4 + 4 + 16        # dimensionality, reserved, permutation, reserved, dims
            t, used = _parse_datatype(b, p)
            types.append(t)
            p += used
        if (nmem == 2 and types[0] == types[1] and types[0].kind == "f" and offsets == [0, types[0].itemsize]
                and size == 2 * types[0].itemsize and names[0].lower() in ("r", "real", "re")
                and names[1].lower() in ("i", "imag", "im")):
            return np.dtype(f"{types[0].byteorder if types[0].byteorder != '=' else '<'}c{size}"), p - off
        return np.dtype({"names": names, "formats": types, "offsets": offsets, "itemsize": size}), p - off
    raise NotImplementedError(f"HDF5 datatype class {cls} is not supported")


# ------------------------------------------------------------------ objects
class _Object:
    """Parsed object header: list of (type, data offset, size) messages."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = []
        r, b = f.r, f.buf
        if b[addr:addr + 4] == b"OHDR":
            self._parse_v2(addr)
        else:
            if b[addr] != 1:
                raise H5Error(f"unknown object header version {b[addr]} at {addr}")
            nmsg = r.u(addr + 2, 2)
            hsize = r.u(addr + 8, 4)
            self._parse_v1_block(addr + 16, hsize, nmsg)

    def _parse_v1_block(self, start, size, budget):
        r = self.f.r
        p, end = start, start + size
        todo = []
        while p + 8 <= end and len(self.msgs) + len(todo) < budget + 64:
            mtype, msize = r.u(p, 2), r.u(p + 2, 2)
            data = p + 8
            if mtype == 0x0010:
                todo.append((r.offs(data), r.lens(data + r.so)))
            self.msgs.append((mtype, data, msize))
            p = data + msize
        for (o, ln) in todo:
            self._parse_v1_block(o, ln, budget)

    def _parse_v2(self, addr):
        r, b = self.f.r, self.f.buf
        flags = b[addr + 5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nb = 1 << (flags & 3)
        csize = r.u(p, nb)
        p += nb
        self._parse_v2_block(p, p + csize, flags)

    def _parse_v2_block(self, p, end, flags):
        r = self.f.r
        todo = []
        while p + 4 <= end:
            mtype, msize, _mflags = self.f.buf[p], r.u(p + 1, 2), self.f.buf[p + 3]
            p += 4
            if flags & 0x04:
                p += 2
            if p + msize > end:
                break
            if mtype == 0x10:
                todo.append((r.offs(p), r.lens(p + r.so)))
            self.msgs.append((mtype, p, msize))
            p += msize
        for (o, ln) in todo:
            if self.f.buf[o:o + 4] != b"OCHK":
                raise H5Error("bad object header continuation block")
            self._parse_v2_block(o + 4, o + ln - 4, flags)

    def find(self, mtype):
        return [(d, s) for (t, d, s) in self.msgs if t == mtype]


class Group:
    def __init__(self, f, obj, name="/"):
        self.f, self.obj, self.name = f, obj, name
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        f, r = self.f, self.f.r
        links = {}
        st = self.obj.find(0x0011)
        if st:
            d, _ = st[0]
            self._walk_btree(r.offs(d), r.offs(d + r.so), links)
        for (d, _s) in self.obj.find(0x0006):           # compact link messages
            b = f.buf
            flags = b[d + 1]
            p = d + 2
            ltype = 0
            if flags & 0x08:
                ltype = b[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            nb = 1 << (flags & 3)
            ln = r.u(p, nb)
            p += nb
            nm = bytes(b[p:p + ln]).decode("utf-8", "replace")
            p += ln
            if ltype == 0:
                links[nm] = r.offs(p)
        for (d, _s) in self.obj.find(0x0002):           # link info: dense storage?
            b = f.buf
            flags = b[d + 1]
            p = d + 2 + (8 if flags & 1 else 0)
            if r.offs(p) != (_UNDEF >> (64 - 8 * r.so)):
                raise NotImplementedError("HDF5 groups with dense link storage are not supported")
        self._links = links

    def _walk_btree(self, addr, heap_addr, links):
        f, r, b = self.f, self.f.r, self.f.buf
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap")
        heap_data = r.offs(heap_addr + 8 + 2 * r.sl)
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise H5Error("bad group B-tree node")
        level, used = b[addr + 5], r.u(addr + 6, 2)
        p = addr + 8 + 2 * r.so
        for i in range(used):
            child = r.offs(p + r.sl)
            p += r.sl + r.so
            if level > 0:
                self._walk_btree(child, heap_addr, links)
                continue
            if b[child:child + 4] != b"SNOD":
                raise H5Error("bad symbol table node")
            nsym = r.u(child + 6, 2)
            q = child + 8
            for _ in range(nsym):
                name_off, ohdr = r.offs(q), r.offs(q + r.so)
                links[r.cstr(heap_data + name_off)] = ohdr
                q += 2 * r.so + 4 + 4 + 16

    @_guard
    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    @_guard
    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node._load()
            if part not in node._links:
                raise KeyError(f"{path!r}: no member {part!r} in {node.name!r}")
            obj = _Object(node.f, node._links[part])
            child_name = node.name.rstrip("/") + "/" + part
            node = Dataset(node.f, obj, child_name) if obj.find(0x0008) else Group(node.f, obj, child_name)
        return node


class Dataset:
    def __init__(self, f, obj, name):
        self.f, self.obj, self.name = f, obj, name
        r, b = f.r, f.buf
        d, _ = obj.find(0x0001)[0]
        ver, rank, flags = b[d], b[d + 1], b[d + 2]
        p = d + (8 if ver == 1 else 4)
        self.shape = tuple(r.lens(p + i * r.sl) for i in range(rank))
        d, _ = obj.find(0x0003)[0]
        self.dtype, _ = _parse_datatype(b, d)
        self.filters = []
        for (d, _s) in obj.find(0x000B):
            ver, nf = b[d], b[d + 1]
            p = d + (8 if ver == 1 else 2)
            for _ in range(nf):
                fid = r.u(p, 2)
                p += 2
                nlen = 0
                if ver == 1 or fid >= 256:
                    nlen = r.u(p, 2)
                    p += 2
                p += 2                                   # flags
                ncd = r.u(p, 2)
                p += 2
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                cd = [r.u(p + 4 * i, 4) for i in range(ncd)]
                p += 4 * ncd
                if ver == 1 and ncd % 2:
                    p += 4
                self.filters.append((fid, cd))
        d, _ = obj.find(0x0008)[0]
        ver = b[d]
        if ver == 3:
            cls = b[d + 1]
            if cls == 0:
                self.layout = ("compact", d + 4, r.u(d + 2, 2))
            elif cls == 1:
                self.layout = ("contiguous", r.offs(d + 2), r.lens(d + 2 + r.so))
            elif cls == 2:
                nd = b[d + 2]
                addr = r.offs(d + 3)
                dims = [r.u(d + 3 + r.so + 4 * i, 4) for i in range(nd)]
                self.layout = ("chunked", addr, tuple(dims[:-1]))
            else:
                raise NotImplementedError(f"HDF5 layout class {cls}")
        elif ver in (1, 2):
            nd, cls = b[d + 1], b[d + 2]
            p = d + 8
            addr = None
            if cls != 0:
                addr = r.offs(p)
                p += r.so
            dims = [r.u(p + 4 * i, 4) for i in range(nd)]
            p += 4 * nd
            if cls == 1:
                self.layout = ("contiguous", addr, int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize)
            elif cls == 2:
                self.layout = ("chunked", addr, tuple(dims[:-1]))
            else:
                size = r.u(p, 4)
                self.layout = ("compact", p + 4, size)
        else:
            raise NotImplementedError(f"HDF5 data layout message version {ver} (libver='latest' files)")

    def __len__(self):
        return self.shape[0]

    def _defilter(self, raw, mask, nbytes):
        for idx in range(len(self.filters) - 1, -1, -1):
            if mask & (1 << idx):
                continue
            fid, cd = self.filters[idx]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                raw = _unshuffle(raw, cd[0] if cd else self.dtype.itemsize)
            elif fid == 3:
                raw = raw[:-4]
            elif fid == 32000:
                raw = lzf_decompress(raw, cd[2] if len(cd) > 2 else nbytes)
            else:
                raise NotImplementedError(f"HDF5 filter {fid} is not supported")
        return raw

    def _chunks(self, addr, rank):
        """Yield (offsets, address, size, filter mask) of every stored chunk."""
        r, b = self.f.r, self.f.buf
        if addr == (_UNDEF >> (64 - 8 * r.so)):
            return
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise H5Error("bad chunk B-tree node")
        level, used = b[addr + 5], r.u(addr + 6, 2)
        p = addr + 8 + 2 * r.so
        ksz = 8 + 8 * (rank + 1)
        for _ in range(used):
            size, mask = r.u(p, 4), r.u(p + 4, 4)
            offs = tuple(r.u(p + 8 + 8 * i, 8) for i in range(rank))
            child = r.offs(p + ksz)
            p += ksz + r.so
            if level > 0:
                yield from self._chunks(child, rank)
            else:
                yield offs, child, size, mask

    @_guard
    def read(self):
        """The whole dataset as a numpy array (native byte order)."""
        kind = self.layout[0]
        n = int(np.prod(self.shape, dtype=np.int64))
        isz = self.dtype.itemsize
        buf = self.f.buf
        if kind in ("compact", "contiguous"):
            addr = self.layout[1]
            if kind == "contiguous" and addr == (_UNDEF >> (64 - 8 * self.f.r.so)):
                out = np.zeros(self.shape, self.dtype)            # never written
            else:
                out = np.frombuffer(buf, dtype=self.dtype, count=n, offset=addr).reshape(self.shape).copy()
        else:
            cshape = self.layout[2]
            rank = len(self.shape)
            out = np.zeros(self.shape, self.dtype)
            cbytes = int(np.prod(cshape, dtype=np.int64)) * isz
            for offs, addr, size, mask in self._chunks(self.layout[1], rank):
                raw = bytes(buf[addr:addr + size])
                if self.filters:
                    raw = self._defilter(raw, mask, cbytes)
                chunk = np.frombuffer(raw, dtype=self.dtype, count=cbytes // isz).reshape(cshape)
                sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, self.shape))
                sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
                out[sl_out] = chunk[sl_in]
        if out.dtype.byteorder == ">":
            out = out.astype(out.dtype.newbyteorder("="))
        return out

    def __getitem__(self, key):
        return self.read()[key]


class File(Group):
    """``File(path)['Header/freq_array'].read()`` -- read-only, whole-dataset reads."""

    def __init__(self, path):
        self._fh = open(path, "rb")
        try:
            self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise H5Error(f"{path}: empty file")
        base = 0
        while True:
            if self.buf[base:base + 8] == _SIG:
                break
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.buf):
                self.close()
                raise H5Error(f"{path}: not an HDF5 file")
        b = self.buf
        ver = b[base + 8]
        if ver in (0, 1):
            so, sl = b[base + 13], b[base + 14]
            self.r = _Reader(b, so, sl)
            p = base + 24 + (4 if ver == 1 else 0)
            p += 4 * so                                  # base, free space, EOF, driver info
            root_ohdr = self.r.offs(p + so)              # symbol table entry: name offset, header addr
        elif ver in (2, 3):
            so, sl = b[base + 9], b[base + 10]
            self.r = _Reader(b, so, sl)
            root_ohdr = self.r.offs(base + 12 + 3 * so)
        else:
            self.close()
            raise H5Error(f"{path}: unknown superblock version {ver}")
        self.f = self
        try:
            Group.__init__(self, self, _Object(self, root_ohdr), "/")
        except _LOW_LEVEL as e:
            self.close()
            raise H5Error(f"{path}: corrupt root group ({type(e).__name__}: {e})") from e

    def close(self):
        try:
            if getattr(self, "buf", None) is not None:
                self.buf.close()
        finally:
            self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
