"""Writes tests/golden/mini.uvh5 (+ mini_uvh5_expected.npz) with h5py, laid out like the files
pyuvdata writes (reference test_data/vis-eor-fgs.uvh5: chunked compound visdata, LZF-compressed
enum flags and float nsamples, contiguous header arrays), plus a few datasets that exercise the
other filters of hydra_pspec_amd.h5lite.  Run with an interpreter that has h5py, e.g.

    /opt/conda/bin/python3.9 tests/golden/make_uvh5_fixture.py
"""
import os
import numpy as np
import h5py

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2024)
Nants, Ntimes, Nfreqs, Npols = 3, 5, 12, 4
pairs = [(0, 1), (0, 2), (2, 1), (0, 0)]      # (2, 1) is stored reversed: the reader must conjugate it
Nbls = len(pairs)
Nblts = Ntimes * Nbls
a1 = np.tile([p[0] for p in pairs], Ntimes).astype(np.int64)      # time-major, like pyuvdata
a2 = np.tile([p[1] for p in pairs], Ntimes).astype(np.int64)
times = np.repeat(2459999.0 + np.arange(Ntimes) / 1000.0, Nbls)
vis = rng.standard_normal((Nblts, Nfreqs, Npols)) + 1j * rng.standard_normal((Nblts, Nfreqs, Npols))
flags = rng.random((Nblts, Nfreqs, Npols)) < 0.2
nsamples = np.ones((Nblts, Nfreqs, Npols), dtype=np.float32)
freqs = 100e6 + 0.5e6 * np.arange(Nfreqs)
pols = np.array([-5, -6, -7, -8], dtype=np.int64)

path = os.path.join(here, "mini.uvh5")
with h5py.File(path, "w") as f:
    h = f.create_group("Header")
    h["Nblts"], h["Nfreqs"], h["Npols"], h["Ntimes"], h["Nbls"] = Nblts, Nfreqs, Npols, Ntimes, Nbls
    h["ant_1_array"], h["ant_2_array"], h["time_array"] = a1, a2, times
    h["freq_array"], h["polarization_array"] = freqs, pols
    h["telescope_name"] = np.bytes_("mini")
    h["version"] = np.bytes_("1.0")
    d = f.create_group("Data")
    cdt = np.dtype([("r", "<f8"), ("i", "<f8")])
    v = np.empty(vis.shape, dtype=cdt)
    v["r"], v["i"] = vis.real, vis.imag
    d.create_dataset("visdata", data=v, chunks=(7, 5, 1))
    d.create_dataset("flags", data=flags, chunks=(10, 6, 2), compression="lzf")
    d.create_dataset("nsamples", data=nsamples, chunks=(5, 3, 2), compression="lzf")
    x = f.create_group("Extra")                       # other layouts / filters
    big = np.arange(40 * 33, dtype=np.int32).reshape(40, 33)
    x.create_dataset("gzip_shuffle", data=big, chunks=(16, 8), compression="gzip", shuffle=True)
    x.create_dataset("fletcher", data=big.astype(">f4"), chunks=(13, 33), fletcher32=True)
    x.create_dataset("contig", data=rng.standard_normal((6, 7)))
    x.create_dataset("c64", data=(vis[:4, :3, 0]).astype(np.complex64))
    x.create_dataset("noise_lzf", data=rng.standard_normal((50, 20)), chunks=(50, 20), compression="lzf")
    many = x.create_group("many")                     # a group large enough to split B-tree nodes
    for k in range(40):
        many[f"item_{k:03d}"] = np.int64(k)
np.savez_compressed(os.path.join(here, "mini_uvh5_expected.npz"), vis=vis, flags=flags, nsamples=nsamples,
                    a1=a1, a2=a2, times=times, freqs=freqs, pols=pols, big=big,
                    contig=np.array(h5py.File(path)["Extra/contig"]), c64=np.array(h5py.File(path)["Extra/c64"]),
                    noise=np.array(h5py.File(path)["Extra/noise_lzf"]))
print("wrote", path, os.path.getsize(path), "bytes")
# a file written with the newest format features (version-4 chunk index): must be refused cleanly
with h5py.File(os.path.join(here, "mini_latest.h5"), "w", libver="latest") as f:
    f.create_dataset("x", data=np.arange(64.0).reshape(8, 8), chunks=(4, 4))
