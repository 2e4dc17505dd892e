"""The C restatement (oracle/csrc/oracle_icp.c) against the golden vectors and the NumPy oracle."""
import numpy as np

from conftest import golden, split
from oracle import c_oracle, icp_oracle


def test_c_oracle_planar_pairs_match_reference():
    g = golden("icp_planar.npz")
    srcs, corrs = split(g["src"], g["src_off"]), split(g["corr"], g["src_off"])
    for k, src in enumerate(srcs):
        T, err, corr, it = c_oracle.run_icp(src, g["tgt"], g["init"][k], epsilon=float(g["epsilon"]))
        assert it == g["iterations"][k]
        np.testing.assert_allclose(T, g["T"][k], atol=1e-9)
        np.testing.assert_allclose(err, g["error"][k], rtol=1e-9)
        np.testing.assert_array_equal(corr, corrs[k])


def test_c_oracle_3d_and_normalized():
    g = golden("icp_3d.npz")
    for k in range(g["init"].shape[0]):
        T, err, corr, it = c_oracle.run_icp(g["src"], g["tgt"], g["init"][k], epsilon=0.001, normalize_error=True)
        assert it == g["iterations"][k]
        np.testing.assert_allclose(T, g["T"][k], atol=1e-8)
        np.testing.assert_allclose(err, g["error"][k], rtol=1e-8)
        np.testing.assert_array_equal(corr, g["corr"][k])


def test_c_oracle_kabsch_cases():
    g = golden("kabsch.npz")
    for k, (p, q) in enumerate(zip(split(g["p"], g["off"]), split(g["q"], g["off"]))):
        np.testing.assert_allclose(c_oracle.kabsch(p, q), g["T"][k], atol=1e-9)


def test_device_recipe_c_vs_numpy_emulation():
    g = golden("icp_planar.npz")
    for k, src in enumerate(split(g["src"], g["src_off"])[:3]):
        T, err, corr, it = c_oracle.run_icp(src, g["tgt"], g["init"][k], epsilon=float(g["epsilon"]), recipe="device")
        r = icp_oracle.run_icp_full(src, g["tgt"], g["init"][k], epsilon=float(g["epsilon"]), recipe="device")
        assert it == r.iterations
        np.testing.assert_array_equal(corr, r.correspondences)
        np.testing.assert_allclose(T, r.T, atol=1e-11)


def test_nn_f32_exact_against_numpy_emulation():
    rng = np.random.default_rng(3)
    for d in (2, 3):
        p = rng.uniform(-20, 20, (500, d)).astype(np.float32)
        q = rng.uniform(-20, 20, (3000, d)).astype(np.float32)
        idx, d2 = c_oracle.nn_f32(p, q)
        np.testing.assert_array_equal(idx, icp_oracle.nn_bruteforce_f32(p, q))
