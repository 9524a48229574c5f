"""UVH5 input without pyuvdata / h5py: the package's HDF5 reader against a
fixture written by h5py (tests/golden/make_uvh5_fixture.py) laid out like the reference's
test_data/vis-eor-fgs.uvh5 (chunked compound visdata, LZF-compressed flags / nsamples)."""
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def expected():
    return np.load(GOLD / "mini_uvh5_expected.npz")


def test_h5lite_layouts_and_filters(expected):
    from hydra_pspec_amd import h5lite
    with h5lite.File(GOLD / "mini.uvh5") as f:
        assert f.keys() == ["Data", "Extra", "Header"]
        v = f["Data/visdata"]
        assert v.shape == expected["vis"].shape and v.dtype == np.complex128 and v.layout[0] == "chunked"
        assert np.array_equal(v.read(), expected["vis"])                      # compound {r, i} -> complex
        fl = f["Data/flags"]
        assert fl.filters[0][0] == 32000                                       # LZF
        assert np.array_equal(fl.read().astype(bool), expected["flags"])      # enum -> int8
        assert np.array_equal(f["Data/nsamples"].read(), expected["nsamples"])
        assert np.array_equal(f["Extra/gzip_shuffle"].read(), expected["big"])
        assert np.array_equal(f["Extra/fletcher"].read(), expected["big"].astype("f4"))   # big-endian source
        assert np.array_equal(f["Extra/contig"].read(), expected["contig"])
        assert f["Extra/c64"].dtype == np.complex64 and np.array_equal(f["Extra/c64"].read(), expected["c64"])
        assert np.array_equal(f["Extra/noise_lzf"].read(), expected["noise"])  # incompressible chunk
        many = f["Extra/many"]
        assert len(many.keys()) == 40 and int(many["item_017"].read()) == 17
        assert int(f["Header/Nblts"].read()) == 20 and f["Header/telescope_name"].read() == b"mini"
        assert "Header/freq_array" in f and "Header/nope" not in f
        with pytest.raises(KeyError):
            f["Data/missing"]


def test_h5lite_refuses_latest_format():
    from hydra_pspec_amd import h5lite
    with h5lite.File(GOLD / "mini_latest.h5") as f:
        with pytest.raises(NotImplementedError):
            f["x"].read()
    with pytest.raises(h5lite.H5Error):
        h5lite.File(GOLD / "small.npz")


def test_lzf_decoder_errors():
    from hydra_pspec_amd import h5lite
    assert h5lite.lzf_decompress(bytes([2, 97, 98, 99]), 3) == b"abc"
    # literal "ab" then a back reference of length 2+2 at distance 2 -> "ababab"
    assert h5lite.lzf_decompress(bytes([1, 97, 98, (2 << 5) | 0, 1]), 6) == b"ababab"
    with pytest.raises(h5lite.H5Error):
        h5lite.lzf_decompress(bytes([1, 97, 98, (2 << 5) | 0, 9]), 6)          # reference before start
    with pytest.raises(h5lite.H5Error):
        h5lite.lzf_decompress(bytes([2, 97, 98, 99]), 5)                       # short output


def test_uvh5_baselines(expected):
    """Antpair order, XX+YY pseudo-Stokes I, XX flags, conjugation to ant1 < ant2
    (run-hydra-pspec.py:316-322, 368-392; utils.py:105-132)."""
    from hydra_pspec_amd import uvh5
    vis, flags, a1, a2 = expected["vis"], expected["flags"], expected["a1"], expected["a2"]
    with uvh5.UVH5File(GOLD / "mini.uvh5") as u:
        assert u.antpairs() == [(0, 0), (0, 1), (0, 2), (1, 2)]
        assert u.antpairs("cross") == [(0, 1), (0, 2), (1, 2)] and u.antpairs("auto") == [(0, 0)]
        with pytest.raises(NotImplementedError):
            u.antpairs("1_2")
        v, f = u.read_baselines([(0, 2), (1, 2)])
    rows02 = np.nonzero((a1 == 0) & (a2 == 2))[0]
    rows21 = np.nonzero((a1 == 2) & (a2 == 1))[0]
    assert v.shape == (2, 5, 12) and f.dtype == bool
    assert np.array_equal(v[0], vis[rows02, :, 0] + vis[rows02, :, 1])
    assert np.array_equal(v[1], np.conj(vis[rows21, :, 0] + vis[rows21, :, 1]))    # stored as (2, 1)
    assert np.array_equal(f[0], flags[rows02, :, 0]) and np.array_equal(f[1], flags[rows21, :, 0])


def test_read_block_and_frequency_selection(expected):
    from hydra_pspec_amd import uvh5
    pairs, vis, flags, ntot, freqs = uvh5.read_uvh5_block(GOLD / "mini.uvh5", 1, 3, freq_range="101-103.2",
                                                         ant_str="cross")
    assert pairs == [(0, 2), (1, 2)] and ntot == 3
    keep = (expected["freqs"] >= 101e6) & (expected["freqs"] <= 103.2e6)
    assert np.array_equal(freqs, expected["freqs"][keep]) and vis.shape == (2, 5, int(keep.sum()))
    full = uvh5.read_uvh5_block(GOLD / "mini.uvh5", 1, 3, ant_str="cross")[1]
    assert np.array_equal(vis, full[:, :, keep])
    f = expected["freqs"] / 1e6
    assert np.array_equal(np.nonzero(uvh5.filter_freqs("100.4,104.9", f))[0], [1, 10])   # closest channels
    assert np.array_equal(np.nonzero(uvh5.filter_freqs("102", f))[0], [4])
    assert not uvh5.filter_freqs("300-400", f).any()
    assert uvh5.read_uvh5_block(GOLD / "mini.uvh5", 7, 9)[0] == []                   # empty block of a rank


def test_h5lite_corrupt_files_fail_cleanly(tmp_path):
    """Random byte corruption must surface as H5Error / KeyError / NotImplementedError -- never as
    a decoder's own exception, a hang or a crash."""
    from hydra_pspec_amd import h5lite
    src = (GOLD / "mini.uvh5").read_bytes()
    rng = np.random.default_rng(0)
    names = ("Data/visdata", "Data/flags", "Header/freq_array", "Extra/gzip_shuffle", "Extra/many/item_003")
    ok = (h5lite.H5Error, KeyError, NotImplementedError)
    for trial in range(150):
        b = bytearray(src)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        p = tmp_path / "fuzz.h5"
        p.write_bytes(b)
        try:
            with h5lite.File(p) as f:
                for name in names:
                    try:
                        f[name].read()
                    except ok:
                        pass
        except ok:
            pass
    (tmp_path / "trunc.h5").write_bytes(src[:5000])
    with pytest.raises(ok):
        with h5lite.File(tmp_path / "trunc.h5") as f:
            f["Data/visdata"].read()
