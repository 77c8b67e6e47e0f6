"""Independent chains of sub-batches: software pipelining of one MarineVecEnv over several HIP streams.

The environments of a batch never interact, so nothing orders lane i's step k+1 behind lane j's step k.  The reference
gets that for free - every `SubprocVecEnv` worker steps on its own (tag_00_Dec2023_simpleControlTurbulence/main_00_sbl.py:145);
a single fused launch per step gives it up again: launch k+1 waits for the LAST wave of launch k (the ramp and tail of
every launch, plus the few microseconds between two dependent launches, leave a tenth of the chip idle).

`ChainStepper` splits the batch into `n_chains` contiguous lane ranges and steps each range on its own stream through
`mvrl_step_range_dev`: chain A's launch k+1 only waits for chain A's launch k, and while it drains and the next one
ramps up, chain B's kernel keeps the SIMDs busy.  With a policy in the loop the same structure overlaps policy(A) with
step(B): `closed_loop(policy, steps)` runs one policy -> step loop per chain.  Results are bit-identical to whole-batch steps (a lane's arithmetic does not depend on the launch geometry).

    stepper = ChainStepper(env, n_chains=2)
    stepper.fork()                        # chains start behind the work already queued on the current stream
    for k in range(K):
        stepper.step(actions[k])          # chain c: lanes [lo_c, hi_c) on stream c; returns the full-batch output tensors
    stepper.join()                        # the current stream waits for every chain
"""
import torch


class ChainStepper:
    def __init__(self, env, n_chains=2, stagger=True, streams=None):
        self.env = env
        n = env.num_envs
        n_chains = max(1, min(int(n_chains), (n + 63) // 64))
        # contiguous ranges whose starts are multiples of 64 lanes (whole waves)
        waves = (n + 63) // 64
        cuts = [min(n, 64 * ((waves * c) // n_chains)) for c in range(n_chains)] + [n]
        self.ranges = [(cuts[c], cuts[c + 1] - cuts[c]) for c in range(n_chains) if cuts[c + 1] > cuts[c]]
        self.n_chains = len(self.ranges)
        dev = torch.device("cuda", env.cfg.device)
        self.streams = streams or [torch.cuda.Stream(device=dev) for _ in range(self.n_chains)]
        self.stagger = bool(stagger) and self.n_chains > 1

    def fork(self):
        """Every chain waits for what is queued on the current stream (actions written there, a reset, ...)."""
        cur = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(cur)

    def phase_delay(self, step_us):
        """Start chain c about c / n_chains of a step late, WITHOUT a cross-stream dependency: a one-wave kernel that spins for
        that long is queued on the chain's stream ahead of its next launch (mvrl_delay_dev).  Chains that start in the same
        instant also end in the same instant, launch after launch, until they happen to drift apart (tens of steps); with
        the offset the second chain's kernel covers the first chain's launch gap and tail from the first step on.  (Making
        chain c wait for an event inside chain 0's first step does the same but costs more than it gains: a cross-stream
        wait takes the command processor 10-20 us.)  Call it with every stream idle (e.g. after a device synchronise) and a
        rough step time in microseconds; long runs do not need it."""
        if not self.stagger or step_us <= 0:
            return
        for c in range(1, self.n_chains):
            self.env.handle.delay_dev(int(step_us * c / self.n_chains), self.streams[c].cuda_stream)

    def step(self, actions, out=None):
        env = self.env
        rt = torch.float64 if env.handle.f64 else torch.float32
        assert actions.is_cuda and actions.is_contiguous() and actions.dtype == rt
        assert tuple(actions.shape) == (env.num_envs, env.action_space.shape[0])
        obs, rew, done = out if out is not None else env._ensure_tensors()
        ptrs = (actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        launch = env.handle.step_range_dev           # enqueues on the given raw stream: no torch stream switch needed
        for (lo, cnt), s in zip(self.ranges, self.streams):
            launch(lo, cnt, *ptrs, s.cuda_stream)
        return obs, rew, done

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)

    def closed_loop(self, policy, steps, actions=None):
        """`steps` env steps with a device-resident policy IN the loop, one loop per chain:

            chain c, on its own stream:   a_c = policy(obs[lo_c:hi_c]);  step lanes [lo_c, hi_c) with a_c;  repeat

        `policy(obs_slice) -> actions_slice` must act row by row (an MLP, the PD / LOS controllers: anything a batch can be
        split for) and is called under `torch.cuda.stream(chain stream)`, so its kernels queue behind the chain's previous env
        step and ahead of its next one.  Step k+1 of a chain really does depend on its step k through the policy - the data
        dependency of every RL roll-out - but not on the OTHER chain: while chain A's policy runs and its next env launch
        ramps up, chain B's env kernel has the chip.  Same observations, actions and states as the joined loop
        `for k: a = policy(obs); obs = env.step_tensors(a)` (bit for bit when the policy is row-wise deterministic).
        Starts from the observations currently in the env's output tensors (reset_tensors() / the previous step); fork() and
        join() around the whole run are done here.  Returns (obs, reward, done, actions) of the last step.

        Measured (tools/policy_loop_bench.py, profiles/r03_policy_loop.txt): with an EAGER torch policy this buys nothing - the
        loop issues every small policy kernel once per chain and becomes bound by Python's launch overhead (C4, elementwise
        policy: 161 vs 160 us per step; 262 144 envs: 75 vs 52) - so use it with policies that are one launch per call (the
        mvrl_policy_* kernels, a captured graph, a compiled module), or not at all: the joined loop is the default for a reason."""
        env = self.env
        obs, rew, done = env._ensure_tensors()
        rt = torch.float64 if env.handle.f64 else torch.float32
        n, ad = env.num_envs, env.action_space.shape[0]
        if actions is None:
            actions = torch.empty((n, ad), dtype=rt, device=obs.device)
        assert actions.is_contiguous() and tuple(actions.shape) == (n, ad) and actions.dtype == rt
        ptrs = (actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        launch = env.handle.step_range_dev
        self.fork()
        for _ in range(int(steps)):
            for (lo, cnt), s in zip(self.ranges, self.streams):
                with torch.cuda.stream(s):
                    actions[lo:lo + cnt].copy_(policy(obs[lo:lo + cnt]))
                launch(lo, cnt, *ptrs, s.cuda_stream)
        self.join()
        return obs, rew, done, actions

    def capture_closed_loop(self, policy, steps, actions=None, joined=False):
        """The closed loop captured ONCE into a HIP graph: `graph.replay()` then runs `steps` env steps with the policy in the loop
        and no host work at all - neither Python's launch overhead, which makes the eager per-chain loops useless
        (profiles/r03_policy_loop.txt), nor the command processor's dependent-launch gap between the small policy kernels.
        One branch of the graph per chain (its own capture stream: policy(A) and step(A) overlap step(B)); joined=True captures
        the single-stream loop `a = policy(obs); obs = step(a)` instead, for comparison.  Replays are exact continuations: the
        env's RNG position lives in its state planes, the policy reads the observation tensors the previous replay left.
        Returns (graph, (obs, reward, done, actions)); the policy must not allocate differently between capture and replay (any
        eager torch module or the mvrl_policy_* kernels)."""
        env = self.env
        obs, rew, done = env._ensure_tensors()
        rt = torch.float64 if env.handle.f64 else torch.float32
        n, ad = env.num_envs, env.action_space.shape[0]
        if actions is None:
            actions = torch.empty((n, ad), dtype=rt, device=obs.device)
        ptrs = (actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        launch = env.handle.step_range_dev
        cap = torch.cuda.Stream(device=obs.device)
        cap.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            cur = torch.cuda.current_stream()
            if joined:
                for _ in range(int(steps)):
                    actions.copy_(policy(obs))
                    env.handle.step_dev(*ptrs, cur.cuda_stream)
            else:
                for s in self.streams:
                    s.wait_stream(cur)
                for _ in range(int(steps)):
                    for (lo, cnt), s in zip(self.ranges, self.streams):
                        with torch.cuda.stream(s):
                            actions[lo:lo + cnt].copy_(policy(obs[lo:lo + cnt]))
                        launch(lo, cnt, *ptrs, s.cuda_stream)
                for s in self.streams:
                    cur.wait_stream(s)
        env.handle.count_launches(0)
        return graph, (obs, rew, done, actions)
