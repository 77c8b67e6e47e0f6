"""Device-resident replay buffer with the reference's symmetry augmentation.

Mirrors `CustomReplayBuffer` (tag_00_Dec2023_simpleControlTurbulence/main_02_sbl_contrib_customBuffer.py:55-160), an SB3
`ReplayBuffer` whose `add` stores every transition five times under mirror/flip sign masks until the buffer has rolled
over more than twice.  Storage is torch tensors on the GPU (`observations`, `next_observations`, `actions`, `rewards`,
`dones`, `timeouts`, shapes `[buffer_size, n_envs, dim]` like SB3's), `add` is one fused HIP kernel fed directly with
the tensors MarineVecEnv.step_tensors returns, `sample` draws uniformly with torch.
"""
import torch

from . import _lib


def plan_add(pos, n_rollovers, buffer_size, n_transforms=5):
    """Slot bookkeeping of one `add` (main_02...py:139-159) without touching data: the reference tests `nRollovers > 2`
    before EACH of the five copies, so a roll-over in the middle of an add stops the synthetic copies right there - the
    slots written are always the first n of the five transforms, starting at `pos` and wrapping.
    Returns (n_written, new_pos, new_n_rollovers, became_full)."""
    n, full = 0, False
    for i in range(n_transforms):
        if n_rollovers > 2 and i != 0:
            continue
        n += 1
        pos += 1
        if pos == buffer_size:
            pos, n_rollovers, full = 0, n_rollovers + 1, True
    return n, pos, n_rollovers, full


class SymmetryReplayBuffer(object):
    N_TRANSFORMS = 5

    def __init__(self, buffer_size, n_envs, obs_dim=11, action_dim=3, device=0, handle_timeouts=False):
        """handle_timeouts=False (default) reproduces the reference pipeline: its envs return info = {} (verySimpleAuv.py:410),
        so `info.get("TimeLimit.truncated", False)` (main_02...py:154) is always False, `timeouts` stays 0 and a time-limit
        `done` is a true terminal for the learner.  True records the time-limit bit of the done byte in `timeouts` (SB3's
        bootstrapping through truncations) - a deliberate deviation, opt-in."""
        if obs_dim != 11 or action_dim != 3:
            raise ValueError("the sign masks are those of the 11-component AuvEnv observation and its 3 actions")
        self.lib = _lib.load()
        self.buffer_size, self.n_envs, self.dev_index = int(buffer_size), int(n_envs), device
        dev = torch.device("cuda", device)
        self.observations = torch.zeros((buffer_size, n_envs, obs_dim), dtype=torch.float32, device=dev)
        self.next_observations = torch.zeros_like(self.observations)
        self.actions = torch.zeros((buffer_size, n_envs, action_dim), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((buffer_size, n_envs), dtype=torch.float32, device=dev)
        self.dones = torch.zeros((buffer_size, n_envs), dtype=torch.uint8, device=dev)
        self.timeouts = torch.zeros((buffer_size, n_envs), dtype=torch.uint8, device=dev)
        self.pos, self.full, self.nRollovers = 0, False, 0
        self.handle_timeouts = bool(handle_timeouts)

    def add(self, obs, next_obs, action, reward, done):
        """obs/next_obs [n_envs, 11], action [n_envs, 3], reward [n_envs] float32 tensors, done [n_envs] uint8 tensor (the
        done bytes of step_tensors: bit 1 marks a time-limit truncation and lands in `timeouts` if handle_timeouts)."""
        n_tr, new_pos, new_roll, became_full = plan_add(self.pos, self.nRollovers, self.buffer_size, self.N_TRANSFORMS)
        for t in (obs, next_obs, action, reward, done):
            assert t.is_cuda and t.is_contiguous()
        _lib.check(self.lib.mvrl_replay_add_sym_dev(
            self.dev_index, obs.data_ptr(), next_obs.data_ptr(), action.data_ptr(), reward.data_ptr(), done.data_ptr(),
            self.n_envs, self.observations.data_ptr(), self.next_observations.data_ptr(), self.actions.data_ptr(),
            self.rewards.data_ptr(), self.dones.data_ptr(), self.timeouts.data_ptr(), self.buffer_size, self.pos, n_tr,
            1 if self.handle_timeouts else 0, torch.cuda.current_stream().cuda_stream))
        self.pos, self.nRollovers = new_pos, new_roll
        self.full = self.full or became_full

    def size(self):
        return self.buffer_size if self.full else self.pos

    def sample(self, batch_size):
        upper = self.size()
        idx = torch.randint(0, upper, (batch_size,), device=self.observations.device)
        env = torch.randint(0, self.n_envs, (batch_size,), device=self.observations.device)
        d = self.dones[idx, env].float() * (1.0 - self.timeouts[idx, env].float())   # SB3: dones * (1 - timeouts)
        return (self.observations[idx, env], self.actions[idx, env], self.next_observations[idx, env], d,
                self.rewards[idx, env])
