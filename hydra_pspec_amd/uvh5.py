"""UVH5 -> per-baseline (Ntimes, Nfreqs) cubes without pyuvdata / h5py / astropy.

Counterpart of the input side of the reference driver (run-hydra-pspec.py:296-322,
368-392; utils.py:105-199): read the file, keep a frequency selection, conjugate
baselines to ``ant1 < ant2`` (``UVData.conjugate_bls`` default), form pseudo-Stokes I
``XX + YY`` (``form_pseudo_stokes_vis``, convention 1.0) and hand out one
``(Ntimes, Nfreqs)`` block per antenna pair in ``get_antpairs()`` order, with the XX
flags.  Every rank / GPU process reads only its own baselines' rows, instead of the
reference's rank-0 load + scatter.
"""
import ast

import numpy as np

from . import h5lite

POL_XX, POL_YY, POL_PI = -5, -6, 1


def filter_freqs(freq_str, freqs_in_mhz):
    """Mask over ``freqs_in_mhz`` selected by ``freq_str`` (reference utils.py:133-199):
    ``'lo-hi'`` keeps lo <= f <= hi, ``'f1,f2,...'`` / ``'f'`` keep the closest channels."""
    f = np.asarray(freqs_in_mhz, dtype=float)
    rng = f"{f.min():.2f} - {f.max():.2f} MHz"
    if "-" in freq_str:
        lo, hi = (float(ast.literal_eval(s)) for s in freq_str.split("-"))
        mask = (f >= lo) & (f <= hi)
        if not mask.any():
            print(f"Frequency range {freq_str} MHz outside of the frequencies in `freqs_in`, {rng}.")
        return mask
    want = [float(ast.literal_eval(s)) for s in freq_str.split(",")]
    bad = [w for w in want if not (f.min() <= w <= f.max())]
    if bad:
        print(f"Frequency(ies) {bad} are not within the range of frequencies in `freqs_in`, {rng}.")
    mask = np.zeros(f.size, dtype=bool)
    for w in want:
        mask[np.argmin(np.abs(f - w))] = True
    return mask


class UVH5File:
    """Header of one UVH5 file and row-wise access to its baselines."""

    def __init__(self, path):
        self.path = str(path)
        self.h5 = h5lite.File(self.path)
        hdr = self.h5["Header"]
        self.ant1 = hdr["ant_1_array"].read().astype(np.int64)
        self.ant2 = hdr["ant_2_array"].read().astype(np.int64)
        self.times = hdr["time_array"].read()
        self.freqs_hz = hdr["freq_array"].read().reshape(-1)          # (1,Nfreqs) in old files
        self.pols = [int(p) for p in hdr["polarization_array"].read().reshape(-1)]
        # UVData.conjugate_bls('ant1<ant2'): rows with ant1 > ant2 are swapped and conjugated
        self.conj = self.ant1 > self.ant2
        lo, hi = np.minimum(self.ant1, self.ant2), np.maximum(self.ant1, self.ant2)
        self.pairs_of_row = np.stack([lo, hi], axis=1)
        self._pairs = sorted(set(map(tuple, self.pairs_of_row.tolist())))
        self._vis = self._flags = None

    def close(self):
        self.h5.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def antpairs(self, ant_str=None):
        """Unique (ant1, ant2) in ``UVData.get_antpairs`` order (ascending baseline number);
        ``ant_str``: None/'all', 'cross' or 'auto' (the forms the reference's YAML uses)."""
        p = self._pairs
        if ant_str in (None, "", "all"):
            return list(p)
        if ant_str == "cross":
            return [q for q in p if q[0] != q[1]]
        if ant_str == "auto":
            return [q for q in p if q[0] == q[1]]
        raise NotImplementedError(f"ant_str={ant_str!r}: only 'all', 'cross' and 'auto' are supported")

    def _load(self):
        if self._vis is None:
            data = self.h5["Data"]
            self._vis = data["visdata"].read()
            self._flags = data["flags"].read().astype(bool)
            if self._vis.ndim == 4:                                   # (Nblts, 1, Nfreqs, Npols) old shape
                self._vis, self._flags = self._vis[:, 0], self._flags[:, 0]

    def read_baselines(self, pairs, freq_mask=None):
        """-> (vis (nbl, Ntimes, Nfreqs) complex128 pseudo-Stokes I, flags (nbl, Ntimes, Nfreqs)
        bool, True = flagged) for the given antenna pairs, times in file order."""
        self._load()
        if POL_PI in self.pols:
            ix, iy = self.pols.index(POL_PI), None
        else:
            if POL_XX not in self.pols or POL_YY not in self.pols:
                raise ValueError("UVH5 file holds neither pI nor both XX and YY polarizations")
            ix, iy = self.pols.index(POL_XX), self.pols.index(POL_YY)
        fsel = slice(None) if freq_mask is None else np.nonzero(freq_mask)[0]
        out_v, out_f = [], []
        for (i, j) in pairs:
            rows = np.nonzero((self.pairs_of_row[:, 0] == i) & (self.pairs_of_row[:, 1] == j))[0]
            if rows.size == 0:
                raise KeyError(f"antenna pair {(i, j)} not in {self.path}")
            rows = rows[np.argsort(self.times[rows], kind="stable")]
            v = self._vis[rows][:, fsel, ix].astype(np.complex128)
            if iy is not None:
                v = v + self._vis[rows][:, fsel, iy]
            cj = self.conj[rows]
            v[cj] = np.conj(v[cj])
            out_v.append(v)
            out_f.append(self._flags[rows][:, fsel, ix])
        return np.array(out_v), np.array(out_f)


def read_uvh5_block(path, lo, hi, freq_range=None, ant_str=None):
    """Baselines ``lo:hi`` (in antpair order) of one UVH5 file.
    Returns (antpairs, vis (nbl,T,N), flags (nbl,T,N) True = flagged, total number of baselines,
    frequencies in Hz of the kept channels)."""
    with UVH5File(path) as u:
        pairs = u.antpairs(ant_str)
        mask = None if not freq_range else filter_freqs(freq_range, u.freqs_hz / 1e6)
        freqs = u.freqs_hz if mask is None else u.freqs_hz[mask]
        sel = pairs[lo:hi]
        if not sel:
            return [], np.zeros((0, 0, freqs.size), complex), np.zeros((0, 0, freqs.size), bool), len(pairs), freqs
        vis, flags = u.read_baselines(sel, mask)
    return sel, vis, flags, len(pairs), freqs
