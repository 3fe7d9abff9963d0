"""`DQN`: the reference's DQN variant (UselessFiles/dqn.py, UselessFiles/replay.py; BASELINE
configs[4]) on the MI355X-native Fly environment.

Same structure and hyper-parameters as the reference (`Net` 3-layer LeakyReLU Q-network, Adam
3e-4, discount 0.99, soft target update tau 0.995, eps = max(0.01, 0.8 - 0.01*step/20), a batch of
128 stored steps x num_envs rows per update).  What the reference does with Python loops runs in
two HIP kernels behind the C ABI: `dqn_eps_greedy` (its per-env argmax loop, dqn.py:94-96) and
`dqn_huber_td` (TD target + Huber loss + the loss gradient at the Q table, dqn.py:68-79).  The
replay is an HBM-resident ring of whole steps (`[capacity, N, .]` tensors) instead of a Python
deque of tuples; a sample is an index_select of 128 step slots.

The upstream file is stale against the current `Fly` and cannot run as written; the two repairs
are explicit:
  D1  `num_obs` is the environment's 73, not the stale default 84 (dqn.py:17);
  D2  `act()` yields ONE scalar per env (dqn.py:89-100) while `Fly.step` takes 18 joint targets:
      the scalar is broadcast to all 18 DoFs.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .fly import Fly
from .params import NUM_DOF


class Net(nn.Module):
    """dqn.py:17-29."""

    def __init__(self, num_obs=73, num_act=18):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(num_obs, 256), nn.LeakyReLU(), nn.Linear(256, 256), nn.LeakyReLU(),
                                 nn.Linear(256, num_act))

    def forward(self, x):
        return self.net(x)


def soft_update(net, net_target, tau):
    """dqn.py:33-36."""
    with torch.no_grad():
        for param_target, param in zip(net_target.parameters(), net.parameters()):
            param_target.data.copy_(param_target.data * tau + param.data * (1.0 - tau))


class ReplayBuffer:
    """replay.py:10-31 as an HBM ring of steps.  `buffer_limit` counts STEPS as upstream, where it is
    1e6 -- impossible with whole vectorised steps (1e6 x 32768 envs x 600 B = 19.7 TB), so the default
    capacity is a STATED number of steps (512: 4 x the 128-step sample; 10.1 GB at 32768 envs, 3.5 % of
    the 288 GB), capped by `budget_bytes`."""

    DEFAULT_STEPS = 512

    def __init__(self, num_envs, num_obs, device, buffer_limit=None, budget_bytes=32 << 30):
        per_step = num_envs * (2 * num_obs + 3) * 4
        want = self.DEFAULT_STEPS if buffer_limit is None else int(buffer_limit)
        self.capacity = max(2, min(want, budget_bytes // per_step))
        self.bytes = self.capacity * per_step
        self.num_envs = num_envs
        c, n = self.capacity, num_envs
        self.obs = torch.empty((c, n, num_obs), device=device)
        self.next_obs = torch.empty((c, n, num_obs), device=device)
        self.action = torch.empty((c, n), device=device)
        self.reward = torch.empty((c, n), device=device)
        self.done = torch.empty((c, n), device=device)
        self.head = 0
        self.count = 0
        self._gen = torch.Generator(device=device)
        self._gen.manual_seed(0)

    def push(self, obs, action, reward, next_obs, done):
        h = self.head
        self.obs[h].copy_(obs); self.next_obs[h].copy_(next_obs)
        self.action[h].copy_(action); self.reward[h].copy_(reward); self.done[h].copy_(done)
        self.head = (h + 1) % self.capacity
        self.count = min(self.count + 1, self.capacity)

    def sample(self, mini_batch_size):
        """`mini_batch_size` distinct stored steps, all envs of each (replay.py:18-28).  The row
        shuffle upstream applies afterwards does not change a mean over the whole batch and is skipped."""
        idx = torch.randperm(self.count, device=self.obs.device, generator=self._gen)[:mini_batch_size]
        f = lambda t: t.index_select(0, idx).flatten(0, 1)   # noqa: E731
        return f(self.obs), f(self.action), f(self.reward), f(self.next_obs), f(self.done)

    def size(self):
        return self.count


class DQN:
    def __init__(self, args, env=None):
        self.args = args
        self.env = env if env is not None else Fly(args)                  # dqn.py:44
        dev = self.device = self.env.device
        n = int(args.num_envs)
        self.act_space = 18                                               # dqn.py:47
        self.discount = 0.99
        self.mini_batch_size = int(getattr(args, "dqn_mini_batch_size", 128))
        self.batch_size = n * self.mini_batch_size
        self.tau = 0.995
        self.num_eval_freq = 100
        self.lr = 3e-4
        self.run_step = 1
        self.score = 0
        cap = getattr(args, "replay_steps", None)
        if cap is None:
            cap = max(4 * self.mini_batch_size, 64)                       # stated default: 4 samples' worth of steps
        self.replay = ReplayBuffer(n, self.env.num_obs, dev, buffer_limit=cap,
                                   budget_bytes=int(getattr(args, "replay_bytes", 32 << 30)))
        print("replay capacity: %d steps x %d envs = %.2f GB of HBM" % (self.replay.capacity, n, self.replay.bytes / 1e9))
        self.q = Net(self.env.num_obs, self.act_space).to(dev)            # D1
        self.q_target = Net(self.env.num_obs, self.act_space).to(dev)
        soft_update(self.q, self.q_target, tau=0.0)
        self.q_target.eval()
        self.optimizer = torch.optim.Adam(self.q.parameters(), lr=self.lr)
        self._lib = _lib.load()
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(getattr(args, "seed", 0)))
        self._coin = torch.empty(n, device=dev)
        self._rand = torch.empty(n, device=dev)
        self._score_acc = torch.zeros((), device=dev)
        self.last_loss = None

    def td_loss_and_grad(self, q_table, act, reward, q_next, done_mask):
        """dqn.py:71-78 through `dqn_huber_td`: returns (loss scalar tensor, dLoss/dQ [B, A])."""
        B, A = q_table.shape
        dq = torch.empty_like(q_table)
        parts = torch.empty((B + 255) // 256, device=q_table.device)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(self._lib.dqn_huber_td(p(q_table), p(act), p(reward), p(q_next), p(done_mask),
                                          C.c_float(self.discount), C.c_int(A), C.c_int64(B), p(dq), p(parts),
                                          _lib.stream_ptr()), "dqn_huber_td")
        return parts.sum() / B, dq

    def update(self):
        """dqn.py:64-85."""
        self.optimizer.zero_grad()
        obs, act, reward, next_obs, done_mask = self.replay.sample(self.mini_batch_size)
        q_table = self.q(obs)
        with torch.no_grad():
            q_next = self.q_target(next_obs)
            loss, dq = self.td_loss_and_grad(q_table.detach().contiguous(), act.contiguous(), reward.contiguous(),
                                             q_next.contiguous(), done_mask.contiguous())
        q_table.backward(dq)
        self.optimizer.step()
        soft_update(self.q, self.q_target, self.tau)
        return loss

    def act(self, obs, epsilon=0.0):
        """dqn.py:89-100: one scalar in [-1,1] per env."""
        n = obs.shape[0]
        self._coin.uniform_(generator=self._gen)
        self._rand.uniform_(generator=self._gen)
        with torch.no_grad():
            q_table = self.q(obs).contiguous()
        out = torch.empty(n, device=obs.device)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(self._lib.dqn_eps_greedy(p(q_table), p(self._coin), p(self._rand), C.c_float(epsilon),
                                            C.c_int(self.act_space), p(out), C.c_int64(n), _lib.stream_ptr()),
                   "dqn_eps_greedy")
        return out

    def run(self):
        """dqn.py:102-126."""
        epsilon = max(0.01, 0.8 - 0.01 * (self.run_step / 20))
        obs = self.env.obs_buf.clone()
        action = self.act(obs, epsilon)
        self.env.step(action.unsqueeze(-1).expand(-1, NUM_DOF).contiguous())     # D2
        next_obs, reward, done = self.env.obs_buf.clone(), self.env.reward_buf.clone(), self.env.reset_buf.clone()
        self.env.reset_async()                                                   # dqn.py:110
        self.replay.push(obs, action, reward, next_obs, (1 - done).to(torch.float32))
        if self.replay.size() > self.mini_batch_size:
            loss = self.update()
            self.last_loss = loss
            self._score_acc += reward.mean() / self.num_eval_freq
            if self.run_step % self.num_eval_freq == 0:
                self.score = float(self._score_acc.item()); self._score_acc.zero_()
                print('Steps: {:04d} | Reward {:.04f} | TD Loss {:.04f} Epsilon {:.04f} Buffer {:03d}'
                      .format(self.run_step, self.score, float(loss.item()), epsilon, self.replay.size()))
                self.score = 0
        self.run_step += 1

    def q_parameters(self):
        return list(self.q.parameters())

    def q_values(self, obs):
        with torch.no_grad():
            return self.q(obs).contiguous()

    def exit(self):
        self.env.exit()
