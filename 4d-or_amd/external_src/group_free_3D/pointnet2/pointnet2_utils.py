"""Group-Free-3D's ``pointnet2_utils`` resolves to the shared HIP operator layer:
its native kernels are byte-for-byte the same nine ops as pointnet2_ops
(GF3D/pointnet2/_ext_src vs OPS/_ext-src), and the keyword
extensions of its ``QueryAndGroup`` / ``GroupAll`` (GF3D/pointnet2/pointnet2_utils.py:301-371)
are implemented there."""
from pointnet2_ops.pointnet2_utils import *  # noqa: F401,F403
from pointnet2_ops.pointnet2_utils import (QueryAndGroup, GroupAll, furthest_point_sample,  # noqa: F401
                                           gather_operation, three_nn, three_interpolate,
                                           grouping_operation, ball_query)
