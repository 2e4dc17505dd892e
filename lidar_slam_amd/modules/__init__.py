"""Mirror of the reference's ``modules`` namespace for the hot path:
``icp`` (modules/icp.py), ``ogm`` (modules/ogm.py), ``localization.poses_from_scan_matching``
(modules/localization.py:95-134) and the SE(2)/SE(3) glue of ``utils`` (modules/utils.py:140-240)."""
