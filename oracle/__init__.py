"""CPU oracle for the ICP / occupancy-grid hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lidar_slam_amd/`` may import this
package: it is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by
the ``cpu_baseline`` leg of ``bench.py`` — there as the checker / the timed
CPU baseline, never as the product path.

Parity status: PINNED.  Every function is checked against outputs of the
reference itself (``/root/reference/modules/{icp,ogm,utils,localization}.py``
imported in the build container by ``tests/golden/make_golden.py``; the
resulting vectors are committed under ``tests/golden/``) and against the
reference's own fixtures (``code/pr2_utils.py:88-100`` Bresenham known-answer
rays, ``outputs/*_20.npy`` SE(2) algebra identities).
"""
