"""Entry point, the reference's trainer.py (trainer.py:1-50) on the MI355X-native Fly/PPO.

Same flags (`--sim_device --compute_device_id --graphics_device_id --num_envs --headless --save
--save_path --save_freq --load --load_path --record --record_dir_name
--time_steps_per_recorded_frame --testing`), same seeding and the same run loop.  Additions:
`--rl_device` (accepted alias; the reference uses sim_device for both), `--variant`
(bigGrav = fly.py, lowGrav = flyLowGrav.py), `--reward` (standing | walking), `--max_steps`
(bounded runs; the reference loops until the viewer's E key), `--seed` and `--log_throughput` (env-steps/s on the score line).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N trainer.py ...`;
each rank owns `--num_envs` envs on its own GPU.
"""
import argparse
import random

import torch


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--sim_device', type=str, default="cuda:0", help='Physics Device in PyTorch-like syntax')
    parser.add_argument('--rl_device', type=str, default=None, help='accepted for compatibility; sim_device is used')
    parser.add_argument('--compute_device_id', default=0, type=int)
    parser.add_argument('--graphics_device_id', type=int, default=0, help='Graphics Device ID')
    parser.add_argument('--num_envs', default=1000, type=int)
    parser.add_argument('--headless', default=False)
    parser.add_argument('--save', type=bool, default=False)
    parser.add_argument('--save_path', type=str, default=None)
    parser.add_argument('--save_freq', type=int, default=100)
    parser.add_argument('--load', type=bool, default=False)
    parser.add_argument('--load_path', type=str, default=None)
    parser.add_argument('--record', type=bool, default=False)
    parser.add_argument('--record_dir_name', type=str, default=None)
    parser.add_argument('--time_steps_per_recorded_frame', type=int, default=2)
    parser.add_argument('--testing', type=bool, default=False)
    parser.add_argument('--variant', type=str, default="bigGrav", choices=["bigGrav", "lowGrav"])
    parser.add_argument('--reward', type=str, default="standing", choices=["standing", "walking"])
    parser.add_argument('--max_steps', type=int, default=0, help='stop after this many env steps (0 = run until env.end)')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--gemm', type=str, default=None, choices=["f32", "bf16x3", "f16x2"],
                        help='arithmetic of the MLP GEMMs: f16x2 (default) = bf16x3 for the rollout policy and the critic pass, the '
                             'optimizer-step gradient in the two-term fp16 split with three product terms and per-class '
                             'power-of-two scales (DESIGN.md 3.4c; a step whose values do not fit fp16 is refused on the device '
                             'and redone in bf16x3); bf16x3 = fp32 operands split exactly into three bf16 terms on the bf16 matrix '
                             'pipe, fp32 accumulate, for EVERY GEMM (DESIGN.md 3.4); f32 = v_mfma_f32_32x32x2_f32.  All are held to '
                             'the reference goldens at the fp32 tolerances.  Default: $FLY_GEMM or f16x2')
    parser.add_argument('--log_throughput', action='store_true',
                        help='append env-steps/s (all ranks, host clock, since the previous score line) to the score line '
                             '(ppo.py:257-260 prints the score only; off by default so that stdout stays the reference\'s)')
    parser.add_argument('--dp_mode', type=str, default="grad_allreduce", choices=["grad_allreduce", "param_average"],
                        help='multi-GPU: all-reduce the gradient every optimizer step (reference algorithm on the global '
                             'batch) or average parameters once per PPO update (non-parity)')
    parser.add_argument('--normalize_advantage', action='store_true',
                        help='normalise advantages over the rollout (not in the reference; off by default)')
    args = parser.parse_args(argv)
    if args.save_path is not None:          # trainer.py:27-34
        args.save = True
    if args.load_path is not None:
        args.load = True
    if args.record_dir_name is not None:
        args.record = True
    return args


def main(argv=None):
    args = parse_args(argv)
    from fly_bproject_amd.dist import broadcast_policy, init_from_env
    from fly_bproject_amd.ppo import PPO

    rank, local_rank, world = init_from_env("cuda")
    args.rank, args.world_size = rank, world
    if world > 1:
        args.sim_device = "cuda:%d" % local_rank
    torch.manual_seed(args.seed)            # trainer.py:24-25
    random.seed(args.seed)
    if args.testing:
        print("## Careful you are in testing mode, no Training will take place ##")
    policy = PPO(args)                      # trainer.py:39
    if args.gemm:
        policy.policy.gemm = args.gemm
    broadcast_policy(policy)
    end = False                             # trainer.py:41-44
    while not end:
        end = policy.run()
        if args.max_steps and policy.run_step >= args.max_steps:
            end = True
    policy.save()                           # trainer.py:48-50
    policy.generate_video()
    policy.exit()
    return policy


if __name__ == '__main__':
    main()
