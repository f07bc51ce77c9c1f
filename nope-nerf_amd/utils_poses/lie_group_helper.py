"""SO(3) helpers under the reference's names (utils_poses/lie_group_helper.py:6-81); the exponential lives in model/common.py."""
from scipy.spatial.transform import Rotation

from model.common import Exp, convert3x4_4x4, make_c2w, vec2skew  # noqa: F401  (re-exported)


def SO3_to_quat(R):
    return Rotation.from_matrix(R).as_quat()


def quat_to_SO3(quat):
    return Rotation.from_quat(quat).as_matrix()
