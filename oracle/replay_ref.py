"""numpy restatement of CustomReplayBuffer.add (tag_00_Dec2023_simpleControlTurbulence/main_02_sbl_contrib_customBuffer.py:
76-160).  TEST INFRASTRUCTURE ONLY.

Pinned: tests/golden/g20_replay.npz was written by executing the reference class itself (oracle/gen/gen_golden_replay.py;
stable_baselines3 is not installed, so the SB3 base-class constructor - array allocation only - is stubbed, `add` runs
unmodified); tests/test_replay_cpu.py checks this restatement against it bit for bit, through the roll-over that falls in
the middle of an add.  The GPU kernel is checked against the fixture and, on larger random batches, against this class."""
import numpy as np

T_OBS = np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], [-1, -1, 1, 1, -1, -1, -1, -1, 1, 1, 1], [-1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1],
                  [1, -1, 1, 1, 1, -1, 1, -1, 1, 1, 1], [1, 1, -1, 1, 1, 1, 1, 1, -1, 1, 1]], dtype=np.float64)   # :109-119
T_ACT = np.array([[1, 1, 1], [-1, -1, 1], [-1, 1, 1], [1, -1, 1], [1, 1, -1]], dtype=np.float64)                  # :120-126


class RefBuffer:
    def __init__(self, buffer_size, n_envs):
        self.buffer_size, self.n_envs = buffer_size, n_envs
        self.observations = np.zeros((buffer_size, n_envs, 11), np.float32)
        self.next_observations = np.zeros((buffer_size, n_envs, 11), np.float32)
        self.actions = np.zeros((buffer_size, n_envs, 3), np.float32)
        self.rewards = np.zeros((buffer_size, n_envs), np.float32)
        self.dones = np.zeros((buffer_size, n_envs), np.float32)
        self.timeouts = np.zeros((buffer_size, n_envs), np.float32)
        self.pos, self.full, self.nRollovers = 0, False, 0

    def add(self, obs, next_obs, action, reward, done, truncated):
        for i in range(5):
            if self.nRollovers > 2 and i != 0:     # :143
                continue
            self.observations[self.pos] = np.array(obs) * T_OBS[i]
            self.next_observations[self.pos] = np.array(next_obs) * T_OBS[i]
            self.actions[self.pos] = np.array(action) * T_ACT[i]
            self.rewards[self.pos] = reward
            self.dones[self.pos] = done
            self.timeouts[self.pos] = truncated
            self.pos += 1
            if self.pos == self.buffer_size:
                self.full = True
                self.pos = 0
                self.nRollovers += 1
