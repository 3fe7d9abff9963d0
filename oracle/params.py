"""TEST INFRASTRUCTURE: the oracle's own restatement of the default parameter sets.

Reference constants: fly.py:16-51, :147-167, :220-228 ("bigGrav", the file ppo.py imports) and
flyLowGrav.py (same lines; diff in SURVEY.md §2 #4).  Joint limits: the 18 revolute joints of
assets/nmf_no_limits_limited_Dofs.urdf in file order; pose: assets/pose_default.yaml (deg).
FlyDyn model constants are build-defined.
"""
import ctypes as C
import math

NDOF, NOBS, NLEG, NABD, NCON = 18, 73, 6, 5, 11

DOF_NAMES = [
    "joint_LFCoxa", "joint_LFFemur", "joint_LFTibia",
    "joint_LHCoxa_roll", "joint_LHFemur", "joint_LHTibia",
    "joint_LMCoxa_roll", "joint_LMFemur", "joint_LMTibia",
    "joint_RFCoxa", "joint_RFFemur", "joint_RFTibia",
    "joint_RHCoxa_roll", "joint_RHFemur", "joint_RHTibia",
    "joint_RMCoxa_roll", "joint_RMFemur", "joint_RMTibia",
]
DOF_LOWER = [
    -1.2282643976845713, -4.986930927481532, -2.362989686468837,
    0.6012615998580322, -5.553724929606129, -3.8187837662418334,
    -0.1644733111051202, -3.8856558255692613, -2.5514814160669523,
    -1.2282643976845713, -4.986930927481532, -2.362989686468837,
    -4.120341207989709, -5.553724929606129, -3.8187837662418334,
    -3.843949339634286, -3.8856558255692613, -2.5514814160669523,
]
DOF_UPPER = [
    1.4495346989023457, 1.4560609499793291, 4.222732123265363,
    4.120341207989709, 1.6139985085925022, 6.979499524663906,
    3.843949339634286, 0.2503410005690172, 5.025832418893524,
    1.4495346989023457, 1.4560609499793291, 4.222732123265363,
    -0.6012615998580322, 1.6139985085925022, 6.979499524663906,
    0.1644733111051202, 0.2503410005690172, 5.025832418893524,
]
POSE_DEG = [
    -0.789880258643274, -67.57373506986399, 43.41127909530307,
    137.6147129043548, -89.38329525236054, 65.7965836898687,
    101.88137481370443, -95.95693707631287, 101.0596642359161,
    -0.07589164341686852, -75.10638587459752, 51.350331169288935,
    -139.89327262013938, -75.37435088662505, 65.58923263208715,
    -104.6230995031921, -105.21451777392465, 98.99614155554471,
]


class OrcConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("substeps", C.c_int32), ("reset_after_sim", C.c_int32),
        ("reward_mode", C.c_int32), ("max_episode_length", C.c_int32),
        ("dt", C.c_float), ("gravity", C.c_float),
        ("kp", C.c_float), ("kd", C.c_float), ("effort", C.c_float), ("vmax", C.c_float),
        ("joint_inertia", C.c_float), ("mass", C.c_float), ("inertia", C.c_float * 3),
        ("kc", C.c_float), ("cdamp", C.c_float), ("mu", C.c_float), ("cvisc", C.c_float),
        ("lin_damp", C.c_float), ("ang_damp", C.c_float),
        ("max_lin_vel", C.c_float), ("max_ang_vel", C.c_float),
        ("femur_len", C.c_float), ("tibia_len", C.c_float), ("alpha0", C.c_float), ("beta0", C.c_float),
        ("dof_lo", C.c_float * NDOF), ("dof_hi", C.c_float * NDOF), ("dof_pose", C.c_float * NDOF),
        ("leg_attach", (C.c_float * 3) * NLEG), ("leg_azimuth", C.c_float * NLEG),
        ("leg_sigma", C.c_float * NLEG), ("abdomen_pts", (C.c_float * 3) * NABD),
        ("start_height", C.c_float), ("target", C.c_float * 3),
        ("dof_vel_scale", C.c_float), ("up_weight", C.c_float), ("heading_weight", C.c_float),
        ("actions_cost_scale", C.c_float), ("energy_cost_scale", C.c_float),
        ("joints_at_limit_cost_scale", C.c_float), ("death_cost", C.c_float),
        ("termination_height", C.c_float), ("termination_height_up", C.c_float),
    ]


def default_config(num_envs, variant="bigGrav"):
    c = OrcConfig()
    c.num_envs = num_envs
    c.reward_mode = 0
    c.max_episode_length = 1500
    c.dt = 1.0 / 60.0
    c.kd, c.vmax = 0.1, 1.0
    if variant == "bigGrav":        # fly.py
        c.substeps, c.reset_after_sim, c.gravity = 15, 0, -9.81 * 1000
        c.kp, c.effort, c.mu, c.energy_cost_scale = 70.0, 30.0, 10.0, 0.005
        c.kc, c.cvisc = 60.0, 0.01
    elif variant == "lowGrav":      # flyLowGrav.py
        c.substeps, c.reset_after_sim, c.gravity = 2, 1, -9.81
        c.kp, c.effort, c.mu, c.energy_cost_scale = 1.3, 1e10, 3.0, 1.0
        c.kc, c.cvisc = 0.5, 0.001
    else:
        raise ValueError(variant)
    # FlyDyn, build-defined
    c.joint_inertia = 1e-3
    c.mass = 1e-3
    c.inertia[:] = [6e-4, 8e-4, 1e-3]
    c.cdamp = 0.05
    c.max_lin_vel, c.max_ang_vel = 1000.0, 64.0
    c.lin_damp, c.ang_damp = 0.5, 2.0
    c.femur_len, c.tibia_len, c.alpha0, c.beta0 = 1.1, 1.2, -0.6, -1.1
    c.dof_lo[:] = DOF_LOWER
    c.dof_hi[:] = DOF_UPPER
    c.dof_pose[:] = [math.radians(d) for d in POSE_DEG]
    # legs in DoF order: LF, LH, LM, RF, RH, RM
    attach = [(0.45, 0.35, -0.25), (-0.40, 0.35, -0.25), (0.0, 0.40, -0.30),
              (0.45, -0.35, -0.25), (-0.40, -0.35, -0.25), (0.0, -0.40, -0.30)]
    azim = [0.87, 2.27, 1.5708, -0.87, -2.27, -1.5708]
    sigma = [1.0, 1.0, 1.0, -1.0, 1.0, 1.0]
    for l in range(NLEG):
        c.leg_attach[l][:] = attach[l]
        c.leg_azimuth[l] = azim[l]
        c.leg_sigma[l] = sigma[l]
    for k in range(NABD):
        c.abdomen_pts[k][:] = (-0.7 - 0.3 * k, 0.0, -0.25 - 0.05 * k)
    c.start_height = 2.0
    c.target[:] = [1000.0, 0.0, 0.0]
    c.dof_vel_scale, c.up_weight, c.heading_weight = 0.2, 0.75, 0.5
    c.actions_cost_scale, c.joints_at_limit_cost_scale = 0.005, 0.1
    c.death_cost, c.termination_height, c.termination_height_up = -2.0, 1.1, 6.0
    return c
