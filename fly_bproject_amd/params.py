"""Parameter sets of the Fly task, as plain data.

Task constants are the ones the reference hard-codes (fly.py:16-51, :147-167, :220-228 for the
"bigGrav" file that ppo.py imports; flyLowGrav.py for the "lowGrav" variant).  Joint limits are
the 18 revolute joints of assets/nmf_no_limits_limited_Dofs.urdf in file order (the build's sim
DoF order) and the pose is assets/pose_default.yaml in radians.  The rigid-body model constants
(FlyDyn) are build-defined: the reference's physics is closed-source PhysX.

`FlyParams` is the ctypes image of `FlyConfig` in include/flyhip.h.
"""
import ctypes as C
import math

NUM_DOF, NUM_OBS, NUM_LEGS, NUM_ABDOMEN, NUM_CONTACT, ROOT_DIM = 18, 73, 6, 5, 11, 13

# fly.py:23-25 lists the actuated joints by name; sorted sim indices (fly.py:288) make the policy
# output j drive sim DoF j, which here is URDF file order:
DOF_NAMES = (
    "joint_LFCoxa", "joint_LFFemur", "joint_LFTibia",
    "joint_LHCoxa_roll", "joint_LHFemur", "joint_LHTibia",
    "joint_LMCoxa_roll", "joint_LMFemur", "joint_LMTibia",
    "joint_RFCoxa", "joint_RFFemur", "joint_RFTibia",
    "joint_RHCoxa_roll", "joint_RHFemur", "joint_RHTibia",
    "joint_RMCoxa_roll", "joint_RMFemur", "joint_RMTibia",
)
# tracked contact bodies, fly.py:299-300 (abdomen first, then leg tips in DoF-leg order)
CONTACT_BODIES = ("A1A2", "A3", "A4", "A5", "A6",
                  "LFTarsus5", "LHTarsus5", "LMTarsus5", "RFTarsus5", "RHTarsus5", "RMTarsus5")

_LEG_LIMITS = {   # (lower, upper) rad, URDF <limit>
    "FCoxa": (-1.2282643976845713, 1.4495346989023457),
    "FFemur": (-4.986930927481532, 1.4560609499793291),
    "FTibia": (-2.362989686468837, 4.222732123265363),
    "HCoxa_roll": (0.6012615998580322, 4.120341207989709),
    "HFemur": (-5.553724929606129, 1.6139985085925022),
    "HTibia": (-3.8187837662418334, 6.979499524663906),
    "MCoxa_roll": (-0.1644733111051202, 3.843949339634286),
    "MFemur": (-3.8856558255692613, 0.2503410005690172),
    "MTibia": (-2.5514814160669523, 5.025832418893524),
}
_POSE_DEG = {   # pose_default.yaml
    "joint_LFCoxa": -0.789880258643274, "joint_LFFemur": -67.57373506986399, "joint_LFTibia": 43.41127909530307,
    "joint_LHCoxa_roll": 137.6147129043548, "joint_LHFemur": -89.38329525236054, "joint_LHTibia": 65.7965836898687,
    "joint_LMCoxa_roll": 101.88137481370443, "joint_LMFemur": -95.95693707631287, "joint_LMTibia": 101.0596642359161,
    "joint_RFCoxa": -0.07589164341686852, "joint_RFFemur": -75.10638587459752, "joint_RFTibia": 51.350331169288935,
    "joint_RHCoxa_roll": -139.89327262013938, "joint_RHFemur": -75.37435088662505, "joint_RHTibia": 65.58923263208715,
    "joint_RMCoxa_roll": -104.6230995031921, "joint_RMFemur": -105.21451777392465, "joint_RMTibia": 98.99614155554471,
}


def _limits(name):
    side, key = name[6], name[7:]
    lo, hi = _LEG_LIMITS[key]
    if side == "R" and key.endswith("Coxa_roll"):   # mirrored roll joints on the right side
        lo, hi = -hi, -lo
    return lo, hi


class FlyParams(C.Structure):
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
        ("dof_lo", C.c_float * NUM_DOF), ("dof_hi", C.c_float * NUM_DOF), ("dof_pose", C.c_float * NUM_DOF),
        ("leg_attach", (C.c_float * 3) * NUM_LEGS), ("leg_azimuth", C.c_float * NUM_LEGS),
        ("leg_sigma", C.c_float * NUM_LEGS), ("abdomen_pts", (C.c_float * 3) * NUM_ABDOMEN),
        ("start_height", C.c_float), ("target", C.c_float * 3),
        ("dof_vel_scale", C.c_float), ("up_weight", C.c_float), ("heading_weight", C.c_float),
        ("actions_cost_scale", C.c_float), ("energy_cost_scale", C.c_float),
        ("joints_at_limit_cost_scale", C.c_float), ("death_cost", C.c_float),
        ("termination_height", C.c_float), ("termination_height_up", C.c_float),
    ]


def default_params(num_envs, variant="bigGrav", reward="standing"):
    """variant: "bigGrav" = fly.py (what ppo.py imports), "lowGrav" = flyLowGrav.py.
    reward: "standing" (fly.py:750, active upstream) or "walking" (fly.py:747-748)."""
    p = FlyParams()
    p.num_envs = int(num_envs)
    p.reward_mode = {"standing": 0, "walking": 1}[reward]
    p.max_episode_length = 1500                       # fly.py:34
    p.dt = 1.0 / 60.0                                 # fly.py:16
    p.kd, p.vmax = 0.1, 1.0                           # fly.py:226-227
    if variant == "bigGrav":
        p.substeps, p.reset_after_sim, p.gravity = 15, 0, -9.81 * 1000   # fly.py:151-154, :660
        p.kp, p.effort, p.mu, p.energy_cost_scale = 70.0, 30.0, 10.0, 0.005   # fly.py:225,228,39,47
        p.kc, p.cvisc = 60.0, 0.01
    elif variant == "lowGrav":
        p.substeps, p.reset_after_sim, p.gravity = 2, 1, -9.81          # flyLowGrav.py:148-151, :661-663
        p.kp, p.effort, p.mu, p.energy_cost_scale = 1.3, 1e10, 3.0, 1.0  # flyLowGrav.py:222, URDF effort, :36, :44
        p.kc, p.cvisc = 0.5, 0.001
    else:
        raise ValueError("unknown variant %r (bigGrav | lowGrav)" % (variant,))
    # FlyDyn (build-defined)
    p.joint_inertia, p.mass = 1e-3, 1e-3
    p.inertia[:] = (6e-4, 8e-4, 1e-3)
    p.cdamp, p.lin_damp, p.ang_damp = 0.05, 0.5, 2.0
    p.max_lin_vel, p.max_ang_vel = 1000.0, 64.0      # Isaac Gym AssetOptions defaults (fly.py:195)
    p.femur_len, p.tibia_len, p.alpha0, p.beta0 = 1.1, 1.2, -0.6, -1.1
    for j, name in enumerate(DOF_NAMES):
        p.dof_lo[j], p.dof_hi[j] = _limits(name)
        p.dof_pose[j] = math.radians(_POSE_DEG[name])
    legs = {   # attach point (mm, thorax frame), rest azimuth, coxa sign; legs in DoF order
        "LF": ((0.45, 0.35, -0.25), 0.87, 1.0), "LH": ((-0.40, 0.35, -0.25), 2.27, 1.0),
        "LM": ((0.0, 0.40, -0.30), 1.5708, 1.0), "RF": ((0.45, -0.35, -0.25), -0.87, -1.0),
        "RH": ((-0.40, -0.35, -0.25), -2.27, 1.0), "RM": ((0.0, -0.40, -0.30), -1.5708, 1.0),
    }
    for l, key in enumerate(("LF", "LH", "LM", "RF", "RH", "RM")):
        att, az, sg = legs[key]
        p.leg_attach[l][:] = att
        p.leg_azimuth[l] = az
        p.leg_sigma[l] = sg
    for k in range(NUM_ABDOMEN):
        p.abdomen_pts[k][:] = (-0.7 - 0.3 * k, 0.0, -0.25 - 0.05 * k)
    p.start_height = 2.0                              # fly.py:33
    p.target[:] = (1000.0, 0.0, 0.0)                  # fly.py:134
    p.dof_vel_scale, p.heading_weight, p.up_weight = 0.2, 0.5, 0.75     # fly.py:43-45
    p.actions_cost_scale, p.joints_at_limit_cost_scale = 0.005, 0.1     # fly.py:46,48
    p.death_cost, p.termination_height, p.termination_height_up = -2.0, 1.1, 6.0   # fly.py:49-51
    return p
