"""SO(3) helpers (reference utils_poses/lie_group_helper.py:6-81).  The so(3) exponential and the 3x4 -> 4x4 padding live in
model/common.py (the reference carries two copies, common.py:277-330 and here); this module re-exports them under the
reference's names and adds the quaternion conversions, which stay on scipy as in the reference (:6-24)."""
from scipy.spatial.transform import Rotation

from model.common import Exp, convert3x4_4x4, make_c2w, vec2skew  # noqa: F401  (re-exported)


def SO3_to_quat(R):
    """(N,3,3) | (3,3) rotation matrices -> (N,4) | (4,) quaternions, scalar-last (x, y, z, w)."""
    return Rotation.from_matrix(R).as_quat()


def quat_to_SO3(quat):
    """(N,4) | (4,) scalar-last quaternions -> (N,3,3) | (3,3)."""
    return Rotation.from_quat(quat).as_matrix()
