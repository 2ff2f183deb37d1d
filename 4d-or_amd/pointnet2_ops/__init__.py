"""pointnet2_ops for AMD Instinct MI355X (gfx950).

Drop-in for the ``pointnet2_ops`` package the reference installs from
scene_graph_prediction/pointnet2_dir/pointnet2_ops_lib (setup.py:26-37): put the
directory that contains this package (``4d-or_amd/``) on ``sys.path`` ahead of
any other ``pointnet2_ops`` and ``pointnet2_ops.pointnet2_utils`` /
``pointnet2_ops.pointnet2_modules`` resolve to the HIP implementation.
"""
import pointnet2_ops.pointnet2_utils  # noqa: F401
import pointnet2_ops.pointnet2_modules  # noqa: F401
from pointnet2_ops._version import __version__  # noqa: F401
