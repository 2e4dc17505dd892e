"""SE(2)/SE(3) glue around the device boundary (reference: modules/utils.py:140-240).

These are O(1) float64 conversions on 3×3 / 4×4 matrices that stay on the host on purpose: the
chain product of relative poses must be float64 and sequential (modules/localization.py:127-130).
Same names, arguments and return conventions as the reference."""
import numpy as np


def transform_points(points, T):
    """Apply a 4×4 transform to (N,3) points or a 3×3 transform to (N,2) points
    (modules/utils.py:140-159); any other combination raises ValueError."""
    d = {(4, 4): 3, (3, 3): 2}.get(T.shape)
    if d is None or points.shape[1] != d:
        raise ValueError("Invalid point or transformation matrix shape.")
    # the homogeneous product itself, as the reference forms it: the translation enters the same dot product
    # as the rotation (p·Rᵀ + t would round differently)
    homogeneous = np.concatenate((points, np.ones((points.shape[0], 1))), axis=1)
    return (T @ homogeneous.T).T[:, :d]


def T_from_pose(pose):
    """(x, y, theta) -> 3×3 homogeneous matrix (modules/utils.py:178-194)."""
    x, y, theta = pose
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, x], [s, c, y], [0.0, 0.0, 1.0]])


def get_relative_pose(pose_t1, pose_t2):
    """inv(T(pose_t1)) @ T(pose_t2) (modules/utils.py:161-176)."""
    return np.dot(np.linalg.inv(T_from_pose(pose_t1)), T_from_pose(pose_t2))


def TSE2_from_TSE3(T_SE3):
    """xy rotation block and xy translation of a 4×4 (modules/utils.py:212-225)."""
    T = np.eye(3)
    T[:2, :2] = T_SE3[:2, :2]
    T[:2, 2] = T_SE3[:2, 3]
    return T


def TSE3_from_TSE2(T_SE2):
    """Embed a 3×3 SE(2) matrix into a 4×4 identity (modules/utils.py:227-240)."""
    T = np.eye(4)
    T[:2, :2] = T_SE2[:2, :2]
    T[:2, 3] = T_SE2[:2, 2]
    return T


def pose_from_T(T):
    """(x, y, atan2(T10, T00)) of a 3×3 or 4×4 matrix (modules/utils.py:196-210)."""
    if T.shape[0] == 4:
        T = TSE2_from_TSE3(T)
    return np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])])
