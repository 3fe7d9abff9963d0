import sys; sys.path.insert(0,".")
import torch
from bench import make_args, _time_launches
from fly_bproject_amd.fly import Fly
for n in (4096, 8192, 16384, 32768, 65536):
    env = Fly(make_args(n))
    a = torch.zeros(n, 18, device="cuda:0").uniform_(-1, 1)
    print(n, "envs: fly_step %.2f us" % (_time_launches(lambda: env.step(a), 200) * 1e6), flush=True)
    env.exit()
