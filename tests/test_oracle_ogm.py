"""The OGM oracle replayed against the reference's Bresenham known answers and golden grids."""
import numpy as np
import pytest

from conftest import golden, split
from oracle.ogm_oracle import (OccupancyGridOracle, bresenham_cells,
                               bresenham_cells_closed_form)


def test_bresenham_known_answers_of_the_reference():
    # code/pr2_utils.py:88-100 (test_bresenham2D): the reference's own two rays
    r1 = bresenham_cells(0, 1, 10, 5)
    assert r1[:, 0].tolist() == list(range(11)) and r1[:, 1].tolist() == [1, 1, 2, 2, 3, 3, 3, 4, 4, 5, 5]
    r2 = bresenham_cells(0, 1, 9, 6)
    assert r2[:, 0].tolist() == list(range(10)) and r2[:, 1].tolist() == [1, 2, 2, 3, 3, 4, 4, 5, 5, 6]


def test_bresenham_golden_rays_both_forms():
    g = golden("bresenham.npz")
    for ray, cells in zip(g["rays"], split(g["cells"], g["off"])):
        np.testing.assert_array_equal(bresenham_cells(*ray), cells)
        np.testing.assert_array_equal(bresenham_cells_closed_form(*ray), cells)


def _grid(g):
    res, maxx, maxy, minx, miny, buf = g["geometry"]
    return OccupancyGridOracle(res, maxx, maxy, minx, miny, buf)


@pytest.mark.parametrize("name", ["ogm_main.npz", "ogm_small.npz"])
def test_cells_counts_and_logodds(name):
    g = golden(name)
    grid = _grid(g)
    assert grid.grid_map.shape == tuple(g["shape"])
    seq = _grid(g)
    for k, pose in enumerate(g["poses"]):
        start, ends = grid.ray_endpoints(pose, g["scan"])
        np.testing.assert_array_equal(start, g["start_cells"][k])        # bit-exact cell indices
        np.testing.assert_array_equal(ends, g["end_cells"][k])
        cnt = grid.scan_counts(pose, g["scan"])
        lo, hi = g["cnt_off"][k], g["cnt_off"][k + 1]
        ref = np.zeros_like(cnt)
        ref[g["cnt_ix"][lo:hi], g["cnt_iy"][lo:hi]] = g["cnt_v"][lo:hi]
        np.testing.assert_array_equal(cnt, ref)                          # bit-exact hit/miss grid
        grid.apply_counts(cnt)
        seq.update_map_sequential(pose, g["scan"])
    ref_lo = np.zeros(grid.grid_map_log_odds.shape, dtype=np.float32)
    ref_lo[g["lo_ix"], g["lo_iy"]] = g["lo_v"]
    np.testing.assert_array_equal(seq.grid_map_log_odds, ref_lo)         # reference arithmetic: exact
    np.testing.assert_allclose(grid.grid_map_log_odds, ref_lo, rtol=0, atol=1e-5)   # count form (O5)
    grid.binarize()
    ref_map = np.unpackbits(g["grid_map"])[: grid.grid_map.size].reshape(grid.grid_map.shape)
    np.testing.assert_array_equal(grid.grid_map, ref_map)


def test_world2grid_conventions():
    grid = OccupancyGridOracle(0.05, 30, 30, -30, -30)
    assert (grid.grid_map_width, grid.grid_map_height) == (1201, 1201)
    assert grid.world2grid(-30.0, -30.0).tolist() == [-1, -1]
    assert grid.world2grid(0.0, 0.0).tolist() == [599, 599]
