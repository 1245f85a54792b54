#!/usr/bin/env python3
"""End-to-end on-device RL smoke: PPO on ReachCube-v0 with N parallel envs, everything resident on one MI355X.

Mirrors the consumer pattern of the reference (examples/gym_manipulation_sb3.py:26-46: SB3 PPO over a vec env with
FilterObservation + FlattenObservation) but without leaving the GPU: observations are zero-copy torch views of the
simulator's state arrays (__cuda_array_interface__), actions are written by the policy straight into the [k][N] device
buffer the step kernel reads.  SB3 itself is not installed in this image, so the PPO update is ~60 lines of torch.

    python examples/ppo_reach_gpu.py --envs 4096 --iters 30
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_lowcostrobot_amd import VecSim  # noqa: E402


class Policy(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden=128):
        super().__init__()
        self.pi = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, act_dim))
        self.v = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, 1))
        self.log_std = nn.Parameter(torch.full((act_dim,), -0.5))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--task", default="reach")
    ap.add_argument("--reward-type", default="dense")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    history = []
    torch.manual_seed(0)
    sim = VecSim(args.task, args.envs, observation_mode="state", reward_type=args.reward_type, base_seed=0)
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
    # zero-copy views, SoA [component][N]; FlattenObservation == concatenate the dict entries
    views = [sim.arm_qpos.torch(), sim.arm_qvel.torch(), sim.cube_pos.torch()] + ([sim.aux_pos.torch()] if sim.aux_pos else [])
    rew_v, done_t, done_u, succ_v = sim.reward.torch(), sim.terminated.torch(), sim.truncated.torch(), sim.is_success.torch()

    def obs_now():
        return torch.cat(views, 0).T.contiguous()  # (N, obs_dim)

    N, k = args.envs, sim.action_dim
    obs_dim = sum(v.shape[0] for v in views)
    pol = Policy(obs_dim, k).to(dev)
    opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
    act_buf = torch.zeros((k, N), device=dev)  # the step kernel reads this [k][N] buffer directly
    gamma, lam, clip = 0.99, 0.95, 0.2
    t_start = time.time()
    for it in range(args.iters):
        O, A, LP, R, D, V = [], [], [], [], [], []
        succ = 0.0
        with torch.no_grad():
            for t in range(args.horizon):
                o = obs_now()
                d = pol.dist(o)
                a = d.sample()
                act_buf.copy_(a.T)  # policy output -> device action buffer (clipped to [-1,1] inside the kernel)
                sim.step_device(act_buf.data_ptr())
                O.append(o); A.append(a); LP.append(d.log_prob(a).sum(-1)); V.append(pol.v(o).squeeze(-1))
                R.append(rew_v.clone()); D.append((done_t | done_u).bool().clone())
                succ += succ_v.float().sum().item()
            last_v = pol.v(obs_now()).squeeze(-1)
            adv = torch.zeros(N, device=dev)
            ADV = [None] * args.horizon
            for t in reversed(range(args.horizon)):
                nv = last_v if t == args.horizon - 1 else V[t + 1]
                nonterm = (~D[t]).float()
                delta = R[t] + gamma * nv * nonterm - V[t]
                adv = delta + gamma * lam * nonterm * adv
                ADV[t] = adv
        O, A, LP, V, ADV = torch.cat(O), torch.cat(A), torch.cat(LP), torch.cat(V), torch.cat(ADV)
        RET = ADV + V
        ADV = (ADV - ADV.mean()) / (ADV.std() + 1e-8)
        idx = torch.randperm(O.shape[0], device=dev)
        for mb in idx.chunk(8):
            d = pol.dist(O[mb])
            ratio = (d.log_prob(A[mb]).sum(-1) - LP[mb]).exp()
            loss_pi = -torch.min(ratio * ADV[mb], ratio.clamp(1 - clip, 1 + clip) * ADV[mb]).mean()
            loss_v = 0.5 * (pol.v(O[mb]).squeeze(-1) - RET[mb]).pow(2).mean()
            opt.zero_grad(); (loss_pi + loss_v).backward(); opt.step()
        mean_r = torch.stack(R).mean().item()
        history.append({"mean_reward": mean_r, "successes": int(succ), "loss": float((loss_pi + loss_v).item())})
        print(f"iter {it:3d}  mean reward/step {mean_r:+.4f}  successes {int(succ):6d}  "
              f"env-steps/s incl. learning {N * args.horizon * (it + 1) / (time.time() - t_start):.3e}", flush=True)
    sim.close()
    return history


if __name__ == "__main__":
    main()
