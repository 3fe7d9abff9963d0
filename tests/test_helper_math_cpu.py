"""CPU: independent checks of the pieces of the parity chain that no reference fixture can pin.

1. `oracle/isaac_stubs` restates the helper formulas of two NVIDIA packages the reference star-imports
   (isaacgym.torch_utils, isaacgymenvs.utils.torch_jit_utils; absent here and unpinned upstream,
   SURVEY.md §8c).  The golden generator runs the reference's TorchScript ON TOP of these stubs, so
   g1's rotation / Euler columns and g3's `scale` inherit them.  Here they are held to
   `scipy.spatial.transform.Rotation` (an implementation nobody in this repo wrote) on random,
   near-upright and gimbal quaternions.  This is an INDEPENDENT CHECK of the stub math, not a
   reference pin: the packages themselves still cannot be run.
2. The joint tables (`fly_bproject_amd/params.py`, `oracle/params.py`) equal the 18 revolute joints of
   the reference's URDF in file order and its pose_default.yaml (skipped where /root/reference does
   not exist, i.e. on the GPU box).
"""
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(REPO, "oracle", "isaac_stubs")
REF = os.environ.get("FLY_REFERENCE", "/root/reference")


@pytest.fixture(scope="module")
def stubs():
    sys.path.insert(0, STUBS)
    try:
        import isaacgym.torch_utils as tu
        import isaacgymenvs.utils.torch_jit_utils as ju
        yield tu, ju
    finally:
        sys.path.remove(STUBS)


def _quats(n, seed):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    q[: n // 4, :3] *= 0.05                                   # near upright
    s = math.sqrt(0.5)
    special = [[0, 0, 0, 1], [0, s, 0, s], [0, -s, 0, s], [s, 0, 0, s], [0, 0, s, s], [1, 0, 0, 0], [0, 0, 1, 0],
               [0.5, 0.5, 0.5, 0.5], [0, 0.7071068, 0, 0.7071067]]
    q[-len(special):] = special
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def test_quaternion_helpers_vs_scipy(stubs):
    from scipy.spatial.transform import Rotation as R
    tu, _ = stubs
    q = _quats(4096, 0)
    rng = np.random.default_rng(1)
    v = rng.normal(size=(len(q), 3)) * 10
    tq, tv = torch.from_numpy(q), torch.from_numpy(v)            # float64: checks the formulas, not rounding
    rot = R.from_quat(q)                                        # scipy: scalar-last xyzw, as Isaac Gym
    np.testing.assert_allclose(tu.quat_rotate(tq, tv).numpy(), rot.apply(v), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tu.quat_rotate_inverse(tq, tv).numpy(), rot.inv().apply(v), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tu.get_basis_vector(tq, tv).numpy(), rot.apply(v), rtol=1e-12, atol=1e-12)
    q2 = _quats(4096, 2)[::-1].copy()
    prod = tu.quat_mul(tq, torch.from_numpy(q2)).numpy()
    ref = (rot * R.from_quat(q2)).as_quat()
    sign = np.sign((prod * ref).sum(1, keepdims=True))           # q and -q are the same rotation
    np.testing.assert_allclose(prod, ref * sign, rtol=1e-12, atol=1e-12)
    conj = tu.quat_conjugate(tq).numpy()
    np.testing.assert_allclose(R.from_quat(conj).apply(v), rot.inv().apply(v), rtol=1e-12, atol=1e-12)
    x = rng.normal(size=(64, 3))
    x[0] = 0.0                                                   # the clamp at 1e-9 keeps a zero vector finite
    nx = tu.normalize(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(nx[1:], x[1:] / np.linalg.norm(x[1:], axis=1, keepdims=True), rtol=1e-14)
    assert np.all(nx[0] == 0.0)


def test_euler_xyz_vs_scipy(stubs):
    """get_euler_xyz returns (roll, pitch, yaw) with R = Rz(yaw) Ry(pitch) Rx(roll), each mod 2 pi:
    scipy's extrinsic 'xyz' angles of the same rotation.  At the gimbal pole (|sin pitch| = 1) roll and
    yaw are not unique, so there the composed rotation is compared instead of the angles."""
    from scipy.spatial.transform import Rotation as R
    tu, _ = stubs
    q = _quats(4096, 3)
    roll, pitch, yaw = (a.numpy() for a in tu.get_euler_xyz(torch.from_numpy(q)))
    for a in (roll, pitch, yaw):
        assert a.min() >= 0.0 and a.max() < 2 * math.pi + 1e-12
    sinp = 2.0 * (q[:, 3] * q[:, 1] - q[:, 2] * q[:, 0])
    reg = np.abs(sinp) < 1 - 1e-6
    assert reg.sum() > 4000 and (~reg).sum() >= 2
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                          # scipy warns about gimbal lock itself
        e = R.from_quat(q).as_euler("xyz")
    d = np.stack([roll, pitch, yaw], 1)[reg] - e[reg]
    d = np.abs((d + math.pi) % (2 * math.pi) - math.pi)
    assert d.max() < 1e-9, d.max()
    back = R.from_euler("xyz", np.stack([roll, pitch, yaw], 1))  # every case, the poles included
    v = np.random.default_rng(4).normal(size=(len(q), 3))
    np.testing.assert_allclose(back.apply(v), R.from_quat(q).apply(v), rtol=0, atol=5e-6)


def test_heading_up_rot_vs_scipy(stubs):
    """compute_heading_and_up / compute_rot (fly.py:789-793) assembled from scipy pieces."""
    from scipy.spatial.transform import Rotation as R
    _, ju = stubs
    n = 1024
    q = _quats(n, 5)
    rng = np.random.default_rng(6)
    pos = rng.normal(size=(n, 3)) * 3
    tgt = np.tile(np.array([1000.0, 0, 0]), (n, 1))
    vel, ang = rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
    to_target = tgt - pos
    to_target[:, 2] = 0
    inv_start = np.tile(np.array([0.0, 0, 0, 1]), (n, 1))
    T = torch.from_numpy
    vec0 = T(np.tile(np.array([1.0, 0, 0]), (n, 1)))
    vec1 = T(np.tile(np.array([0.0, 0, 1]), (n, 1)))
    tq, up_proj, heading_proj, up_vec, heading_vec = ju.compute_heading_and_up(T(q), T(inv_start), T(to_target), vec0, vec1, 2)
    rot = R.from_quat(q)
    np.testing.assert_allclose(up_vec.numpy(), rot.apply([0, 0, 1.0]), atol=1e-12)
    np.testing.assert_allclose(heading_vec.numpy(), rot.apply([1.0, 0, 0]), atol=1e-12)
    np.testing.assert_allclose(up_proj.numpy(), rot.apply([0, 0, 1.0])[:, 2], atol=1e-12)
    dirs = to_target / np.linalg.norm(to_target, axis=1, keepdims=True)
    np.testing.assert_allclose(heading_proj.numpy(), (rot.apply([1.0, 0, 0]) * dirs).sum(1), atol=1e-12)
    vel_loc, angvel_loc, roll, pitch, yaw, angle = ju.compute_rot(tq, T(vel), T(ang), T(tgt), T(pos))
    np.testing.assert_allclose(vel_loc.numpy(), rot.inv().apply(vel), atol=1e-12)
    np.testing.assert_allclose(angvel_loc.numpy(), rot.inv().apply(ang), atol=1e-12)
    # upstream's walk_target_angle uses index 2 (z), not 1: keep it
    np.testing.assert_allclose(angle.numpy(), np.arctan2(tgt[:, 2] - pos[:, 2], tgt[:, 0] - pos[:, 0]) - yaw.numpy(), atol=1e-12)


def test_scale_unscale_are_inverse_affine_maps(stubs):
    tu, _ = stubs
    rng = np.random.default_rng(7)
    lo = torch.from_numpy(rng.uniform(-5, 0, 18)); hi = lo + torch.from_numpy(rng.uniform(0.5, 8, 18))
    x = torch.from_numpy(rng.uniform(-1.5, 1.5, (100, 18)))
    y = tu.scale(x, lo, hi)
    assert torch.allclose(tu.scale(torch.full((1, 18), -1.0, dtype=torch.float64), lo, hi), lo.unsqueeze(0))
    assert torch.allclose(tu.scale(torch.full((1, 18), 1.0, dtype=torch.float64), lo, hi), hi.unsqueeze(0))
    assert torch.allclose(tu.unscale(y, lo, hi), x, atol=1e-12)


def test_oracle_pack_obs_vs_scipy():
    """The C oracle's own obs pack (oracle/fly_oracle.c, the checker of every GPU obs test) against the
    scipy assembly of fly.py:771-805 -- so the checker does not rest on the stubs alone."""
    from scipy.spatial.transform import Rotation as R
    from oracle import oracle as O
    n = 512
    cfg = O.default_config(n)
    q = _quats(n, 8).astype(np.float32)
    rng = np.random.default_rng(9)
    s = O.EnvState(n)
    s.root[:, :3] = rng.normal(0, 2, (n, 3)); s.root[:, 2] = rng.uniform(0.5, 5, n)
    s.root[:, 3:7] = q
    s.root[:, 7:] = rng.normal(0, 2, (n, 6))
    O.pack_obs(cfg, s)
    q64 = s.root[:, 3:7].astype(np.float64)
    rot = R.from_quat(q64)
    np.testing.assert_allclose(s.obs[:, 1:4], rot.inv().apply(s.root[:, 7:10].astype(np.float64)), atol=2e-5)
    np.testing.assert_allclose(s.obs[:, 4:7], rot.inv().apply(s.root[:, 10:13].astype(np.float64)), atol=2e-5)
    np.testing.assert_allclose(s.obs[:, 10], rot.apply([0, 0, 1.0])[:, 2], atol=2e-6)
    sinp = 2.0 * (q64[:, 3] * q64[:, 1] - q64[:, 2] * q64[:, 0])
    reg = np.abs(sinp) < 1 - 1e-4
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        e = rot.as_euler("xyz")
    for col, k in ((8, 0), (66, 1), (7, 2)):                    # roll, pitch, yaw columns of the 73
        d = np.abs((s.obs[reg, col] - e[reg, k] + math.pi) % (2 * math.pi) - math.pi)
        assert d.max() < 2e-3 * max(1.0, 1.0 / np.sqrt(1 - sinp[reg] ** 2).min() * 1e-3), (col, d.max())


# ---------------------------------------------------------------------------------------------------
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "assets")), reason="reference assets only exist in the build container")


def _urdf_revolute_joints():
    text = open(os.path.join(REF, "assets", "nmf_no_limits_limited_Dofs.urdf")).read()
    out = []
    for m in re.finditer(r'<joint name="([^"]+)" type="revolute">(.*?)</joint>', text, flags=re.S):
        lim = re.search(r'<limit[^>]*lower="([^"]+)"[^>]*upper="([^"]+)"', m.group(2))
        out.append((m.group(1), float(lim.group(1)), float(lim.group(2))))
    return out


@needs_ref
def test_joint_tables_equal_the_reference_assets():
    """fly.py:194 loads this URDF, fly.py:61-65 this YAML (degrees); the 18 actuated names are fly.py:23-25."""
    import yaml
    from fly_bproject_amd import params as PP
    from oracle import params as OP
    joints = _urdf_revolute_joints()
    assert len(joints) == 18
    names = [j[0] for j in joints]
    assert list(PP.DOF_NAMES) == names and list(OP.DOF_NAMES) == names             # URDF file order = sim DoF order here
    pose = yaml.safe_load(open(os.path.join(REF, "assets", "pose_default.yaml")))["joints"]
    prod, orc = PP.default_params(4), OP.default_config(4)
    for j, (name, lo, hi) in enumerate(joints):
        for cfg in (prod, orc):
            assert cfg.dof_lo[j] == np.float32(lo) and cfg.dof_hi[j] == np.float32(hi), name
            assert cfg.dof_pose[j] == np.float32(math.radians(pose[name])) or \
                abs(cfg.dof_pose[j] - math.radians(pose[name])) < 1e-7, name
    # the actuated names of the reference source are exactly these joints
    src = open(os.path.join(REF, "fly.py")).read()
    listed = set(re.findall(r'"(joint_[LR][FMH](?:Coxa(?:_roll)?|Femur|Tibia))"', src))
    assert listed == set(names)


@needs_ref
@pytest.mark.parametrize("variant,fname", [("bigGrav", "fly.py"), ("lowGrav", "flyLowGrav.py")])
def test_task_constants_equal_the_reference_source(variant, fname):
    """The hard-coded task constants both parameter tables hand to the kernels / the oracle, read back
    from the text of the reference file they claim to follow (fly.py:16-51, :147-167, :220-228)."""
    from fly_bproject_amd import params as PP
    from oracle import params as OP
    src = open(os.path.join(REF, fname)).read()

    def attr(name):
        m = re.search(r"^\s*self\.%s\s*=\s*([-+0-9.eE/ ]+)" % name, src, flags=re.M)
        assert m, name
        return float(eval(m.group(1), {"__builtins__": {}}))      # plain arithmetic literals such as 1/60 only

    def fill(name):
        m = re.search(r"^\s*dof_props\[['\"]%s['\"]\]\.fill\(([-+0-9.eE]+)\)" % name, src, flags=re.M)
        return float(m.group(1)) if m else None

    g = re.search(r"sim_params\.gravity\s*=\s*gymapi\.Vec3\(0\.0,\s*0\.0,\s*([-+0-9.*eE ]+)\)", src)
    gravity = float(eval(g.group(1), {"__builtins__": {}}))
    substeps = int(re.search(r"sim_params\.substeps\s*=\s*(\d+)", src).group(1))
    for cfg in (PP.default_params(4, variant), OP.default_config(4, variant)):
        assert cfg.substeps == substeps
        assert cfg.gravity == np.float32(gravity)
        assert cfg.dt == np.float32(attr("dt"))
        assert cfg.max_episode_length == int(attr("max_episode_length"))
        assert cfg.mu == np.float32(attr("plane_static_friction")) == np.float32(attr("plane_dynamic_friction"))
        for name in ("dof_vel_scale", "heading_weight", "up_weight", "actions_cost_scale", "energy_cost_scale",
                     "joints_at_limit_cost_scale", "death_cost", "termination_height", "termination_height_up"):
            assert getattr(cfg, name) == np.float32(attr(name)), name
        assert cfg.kp == np.float32(fill("stiffness")) and cfg.kd == np.float32(fill("damping"))
        assert cfg.vmax == np.float32(fill("velocity"))
        if fill("effort") is not None:
            assert cfg.effort == np.float32(fill("effort"))
    # reset ordering: fly.py resets BEFORE simulate(), flyLowGrav.py after it (diff lines 657-663)
    step = src[src.index("def step(self"):]
    assert (step.index("self.reset()") < step.index("self.simulate()")) == (PP.default_params(4, variant).reset_after_sim == 0)
