"""lidar_slam_amd — MI355X (gfx950) native ICP scan matching and occupancy-grid ray casting.

Drop-in for the hot path of fadli0029/LiDAR-SLAM: ``modules/icp.py`` and the per-scan log-odds
ray casting of ``modules/ogm.py``, behind the same Python call surface
(``lidar_slam_amd.modules.icp.run_icp``, ``lidar_slam_amd.modules.ogm.OccupancyGridMap``,
``lidar_slam_amd.modules.localization.poses_from_scan_matching``).  Host code is Python calling
hand-written HIP kernels through the C ABI of ``include/lslam.h`` (ctypes).  There is no CPU
fallback: without ``liblslam_hip.so`` and a HIP device every compute call raises.
"""
from ._lib import LslamError, device_count, lib  # noqa: F401
from .batch import release_thread_resources  # noqa: F401

__version__ = "0.1.0"
