"""Host-side logic that needs no GPU: packing of scan lists (PackedScans keeps the semantics of the plain
list the reference's pre-processing returns, localization.py:252-307 / main.py:84-87)."""

import numpy as np
import pytest

from lidar_slam_amd.batch import PackedScans, pack_clouds


def _scans(rng, sizes):
    off = np.zeros(len(sizes) + 1, np.int64)
    off[1:] = np.cumsum(sizes)
    return PackedScans(rng.normal(size=(int(off[-1]), 2)), off)


def test_packed_scans_is_a_list_of_views_and_packs_without_copying_element_by_element():
    rng = np.random.default_rng(0)
    z = _scans(rng, [5, 0, 7, 3])
    assert isinstance(z, list) and len(z) == 4 and [s.shape for s in z] == [(5, 2), (0, 2), (7, 2), (3, 2)]
    flat, off = pack_clouds(z)
    assert flat.dtype == np.float64 and off is z.off and flat is z.flat        # float64-exact mode: the packed array itself, no copy
    f32, off32 = pack_clouds(z, dtype=np.float32)
    assert f32.dtype == np.float32 and off32 is z.off
    np.testing.assert_array_equal(f32, z.flat.astype(np.float32))
    # writing INTO an element writes into the packed array (elements are views)
    z[2][:] = 1.5
    flat2, _ = pack_clouds(z)
    assert (flat2[off[2]:off[3]] == 1.5).all()


def test_mutating_the_list_is_honoured_like_a_plain_list():
    rng = np.random.default_rng(1)
    mutations = {
        "setitem": lambda z: z.__setitem__(1, z[1][::2]),
        "delitem": lambda z: z.__delitem__(0),
        "reverse": lambda z: z.reverse(),
        "sort": lambda z: z.sort(key=lambda a: a.shape[0]),
        "append": lambda z: z.append(np.ones((2, 2))),
        "insert": lambda z: z.insert(1, np.ones((2, 2))),
        "pop": lambda z: z.pop(),
        "extend": lambda z: z.extend([np.zeros((1, 2))]),
        "iadd": lambda z: z.__iadd__([np.zeros((1, 2))]),
        "slice": lambda z: z.__setitem__(slice(0, 2), [z[1], z[0]]),
    }
    for name, edit in mutations.items():
        z = _scans(rng, [4, 6, 3, 5])
        edit(z)
        assert z.packed() is None, name
        flat, off = pack_clouds(z)                                   # re-packed from the list as it is now
        want = np.concatenate([np.asarray(s) for s in z], axis=0)
        np.testing.assert_array_equal(flat, want, err_msg=name)
        np.testing.assert_array_equal(pack_clouds(z, dtype=np.float32)[0], want.astype(np.float32), err_msg=name)
        assert off.tolist() == np.concatenate(([0], np.cumsum([s.shape[0] for s in z]))).tolist(), name


def test_pack_clouds_rejects_mixed_dimensions():
    import pytest
    with pytest.raises(ValueError):
        pack_clouds([np.zeros((3, 2)), np.zeros((3, 3))])


def test_c_packing_helper_gathers_converts_and_checks_a_scan_list():
    """csrc/pyhost: the list of scans walked through the buffer protocol and copied by several threads — offsets, order,
    float32 <-> float64 conversion, the finiteness verdict, the too-small-destination answer, and its refusals"""
    from lidar_slam_amd import _lib
    ph = _lib.pyhost()
    assert ph is not None, "lidar_slam_amd/lib/_lslam_pyhost.so is missing: make -C lidar_slam_amd/csrc"
    rng = np.random.default_rng(5)
    sizes = [0, 7, 1081, 1, 650, 0, 333] * 40
    z = [rng.normal(size=(n, 2)) for n in sizes]
    total = sum(sizes)
    off = np.empty(len(z) + 1, np.int64)
    for threads in (1, 3, 8):
        for dst_dtype in (np.float64, np.float32):
            dst = np.full((total, 2), -7.0, dst_dtype)
            r = ph.pack_list(z, dst.ctypes.data, total, 2, int(dst_dtype == np.float64), off.ctypes.data, threads)
            assert r == (total, 1, len(z))
            np.testing.assert_array_equal(off, np.concatenate(([0], np.cumsum(sizes))))
            np.testing.assert_array_equal(dst, np.concatenate(z).astype(dst_dtype))
    # a float32 scan among float64 ones: converted, and counted
    z32 = list(z); z32[3] = z32[3].astype(np.float32); z32[2] = z32[2].astype(np.float32)
    dst = np.empty((total, 2))
    assert ph.pack_list(z32, dst.ctypes.data, total, 2, 1, off.ctypes.data, 4) == (total, 1, len(z) - 2)
    np.testing.assert_array_equal(dst, np.concatenate([a.astype(np.float64) for a in z32]))
    # destination too small: nothing copied, the needed size comes back
    assert ph.pack_list(z, dst.ctypes.data, total - 1, 2, 1, off.ctypes.data, 4)[:2] == (total, -1)
    # a non-finite coordinate anywhere
    bad = list(z); bad[2] = bad[2].copy(); bad[2][500, 1] = np.inf
    assert ph.pack_list(bad, dst.ctypes.data, total, 2, 1, off.ctypes.data, 4)[1] == 0
    # what it cannot take: wrong width, a non-contiguous view, an integer array, not a list
    assert ph.pack_list(z + [np.zeros((3, 3))], dst.ctypes.data, total + 3, 2, 1, np.empty(len(z) + 2, np.int64).ctypes.data, 2) is None
    assert ph.pack_list([z[2][:, ::-1]], dst.ctypes.data, total, 2, 1, off.ctypes.data, 2) is None
    assert ph.pack_list([np.zeros((4, 2), np.int32)], dst.ctypes.data, total, 2, 1, off.ctypes.data, 2) is None
    assert ph.pack_list(tuple(z), dst.ctypes.data, total, 2, 1, off.ctypes.data, 2) is None


def test_pack_clouds_keeps_the_precision_it_is_given():
    rng = np.random.default_rng(6)
    z64 = [rng.normal(size=(n, 2)) for n in (5, 9, 2)]
    z32 = [a.astype(np.float32) for a in z64]
    assert pack_clouds(z64)[0].dtype == np.float64 and pack_clouds(z32)[0].dtype == np.float32
    assert pack_clouds([z32[0], z64[1], z32[2]])[0].dtype == np.float64            # one float64 scan keeps the whole batch exact
    assert pack_clouds(z64, dtype=np.float32)[0].dtype == np.float32


def test_tiled_copies_for_multi_start_registration(monkeypatch):
    """icp_warm_up._tiled: every cloud n_starts times in a row — in the pinned staging buffer when the library provides one,
    in an ordinary array otherwise; same contents either way (here: no GPU, the fallback)"""
    from lidar_slam_amd import _lib, icp_warm_up as w
    def no_staging(slot, count, dtype):
        raise RuntimeError("no device")
    monkeypatch.setattr(_lib, "staging_array", no_staging)
    a = np.arange(12.).reshape(4, 3)
    b = np.arange(100., 106.).reshape(2, 3)
    out = w._tiled([a, b], 3, 18, 3, 0)
    np.testing.assert_array_equal(out, np.vstack([np.tile(a, (3, 1)), np.tile(b, (3, 1))]))
    assert out.dtype == a.dtype and out.flags["C_CONTIGUOUS"]
    out32 = w._tiled([a.astype(np.float32)], 2, 8, 3, 1)
    assert out32.dtype == np.float32 and out32.shape == (8, 3)


def test_pack_clouds_accepts_what_the_c_helper_declines():
    """elements that are Python lists, non-contiguous views or integer arrays: the NumPy path normalises them (with or without
    the C helper / a device), same packed result"""
    from lidar_slam_amd.batch import pack_clouds
    base = np.arange(24.0).reshape(12, 2)
    clouds = [[[0.0, 1.0], [2.0, 3.0]], base[::3], np.zeros((0, 2)), np.array([[5, 6]])]
    want = np.concatenate([np.asarray(c, dtype=np.float64).reshape(-1, 2) for c in clouds])
    for staging in (None, 0):
        flat, off = pack_clouds(clouds, staging=staging)
        assert flat.dtype == np.float64 and np.array_equal(flat, want)
        assert off.tolist() == [0, 2, 6, 6, 7]
    flat32, _ = pack_clouds([np.ones((3, 2), np.float32), [[1.0, 2.0]]], dtype=np.float32)
    assert flat32.dtype == np.float32 and flat32.shape == (4, 2)
    with pytest.raises(ValueError):
        pack_clouds([np.ones((3, 2)), np.ones((3, 3))])
