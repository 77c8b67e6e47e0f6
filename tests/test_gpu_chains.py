"""Lane-range stepping (mvrl_step_range_dev) and the chain stepper built on it: a batch stepped as independent chains of
sub-batches on their own streams must end in EXACTLY the state, observations, rewards and dones of whole-batch steps -
random auto-resets included (a lane's arithmetic and its Philox stream do not depend on the launch geometry)."""
import numpy as np
import pytest

from marinevehiclereinforcementlearning_amd import _lib, params as P

pytestmark = pytest.mark.gpu


def _flow():
    from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
    f = ReconstructedFlow.synthetic(n_modes=4, n_time=64)
    f.scale(11., 1., 2., translate=(-1.65, -1.1))
    return f


@pytest.mark.parametrize("model,n,chains", [("rov6", 8192 + 77, 2), ("rov6", 4096, 3), ("rov3", 5000, 2), ("auv", 6000, 2),
                                            ("auv_cyl", 3000, 4)])
def test_chains_bit_identical_to_whole_batch_steps(model, n, chains):
    import torch
    from marinevehiclereinforcementlearning_amd.chains import ChainStepper
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    steps, max_steps = 23, 9   # several random auto-resets inside the run
    kw = dict(seed=11, maxSteps=max_steps, infos="lean")
    flow_a, flow_b = _flow(), _flow()
    ea = MarineVecEnv(model, n, flow=flow_a, **kw)
    eb = MarineVecEnv(model, n, flow=flow_b, **kw)
    act_dim = ea.action_space.shape[0]
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.rand((steps, n, act_dim), device="cuda", generator=g) * 2 - 1
    oa0 = ea.reset_tensors().clone()
    ob0 = eb.reset_tensors().clone()
    assert torch.equal(oa0, ob0)
    st = ChainStepper(eb, n_chains=chains)
    assert st.n_chains == chains and sum(c for _, c in st.ranges) == n and all(lo % 64 == 0 for lo, _ in st.ranges)
    outs_a, outs_b = [], []
    for k in range(steps):
        o, r, d = ea.step_tensors(acts[k])
        outs_a.append((o.clone(), r.clone(), d.clone()))
    st.fork()
    bufs = [tuple(torch.empty_like(t) for t in outs_a[0]) for _ in range(steps)]
    for k in range(steps):
        st.step(acts[k], out=bufs[k])   # own output tensors per step: chains run ahead of each other
    st.join()
    torch.cuda.synchronize()
    for k in range(steps):
        for ta, tb in zip(outs_a[k], bufs[k]):
            assert torch.equal(ta, tb), (model, k)
    assert np.array_equal(ea.get_state(), eb.get_state())
    assert np.array_equal(ea.handle.terminal_obs(), eb.handle.terminal_obs())
    assert int(ea.handle.episode_counter().max()) >= 3     # resets really happened
    ea.close(); eb.close()


@pytest.mark.parametrize("model,n,chains", [("rov6", 8192 + 77, 2), ("rov3", 5000, 3), ("auv", 6000, 2)])
def test_closed_loop_with_a_policy_per_chain(model, n, chains):
    """ChainStepper.closed_loop: a device-resident row-wise policy in the loop, one policy -> step loop per chain on its own
    stream - the same observations, actions and final state, bit for bit, as the joined loop a = policy(obs); obs = step(a),
    random auto-resets included (step k+1 of a chain depends on its step k through the policy, not on the other chain)."""
    import torch
    from marinevehiclereinforcementlearning_amd.chains import ChainStepper
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    steps = 21
    kw = dict(seed=3, maxSteps=8, infos="lean")
    ea, eb = MarineVecEnv(model, n, flow=_flow(), **kw), MarineVecEnv(model, n, flow=_flow(), **kw)
    ad, od = ea.action_space.shape[0], ea.observation_space.shape[0]
    g = torch.Generator(device="cuda").manual_seed(9)
    w = torch.rand((1, ad), device="cuda", generator=g) * 4 - 2
    b = torch.rand((1, ad), device="cuda", generator=g) - 0.5

    def policy(obs):          # elementwise, so row-wise deterministic whatever the slice
        return torch.tanh(obs[:, :ad] * w + obs[:, od - ad:] * 0.5 + b)
    oa = ea.reset_tensors()
    eb.reset_tensors()
    for k in range(steps):
        a_ref = policy(oa)
        oa, ra, da = ea.step_tensors(a_ref.contiguous())
    st = ChainStepper(eb, n_chains=chains)
    ob, rb, db, a_last = st.closed_loop(policy, steps)
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(a_ref, a_last)
    assert np.array_equal(ea.get_state(), eb.get_state())
    assert int(ea.handle.episode_counter().max()) >= 3
    ea.close(); eb.close()


@pytest.mark.parametrize("model,n,joined", [("rov6", 8192 + 77, False), ("rov6", 4096, True), ("auv", 6000, False)])
def test_closed_loop_captured_into_a_hip_graph(model, n, joined):
    """ChainStepper.capture_closed_loop: the policy -> step loop (one graph branch per chain, or the joined single-stream loop)
    captured ONCE and replayed: 3 replays of 7 steps == 21 eager steps of the joined loop, bit for bit, auto-resets included
    (the env's RNG position lives in its state planes, so a replay is the exact continuation)."""
    import torch
    from marinevehiclereinforcementlearning_amd.chains import ChainStepper
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    kw = dict(seed=3, maxSteps=8, infos="lean")
    ea, eb = MarineVecEnv(model, n, flow=_flow(), **kw), MarineVecEnv(model, n, flow=_flow(), **kw)
    ad, od = ea.action_space.shape[0], ea.observation_space.shape[0]
    g = torch.Generator(device="cuda").manual_seed(9)
    w = torch.rand((1, ad), device="cuda", generator=g) * 4 - 2

    def policy(obs):
        return torch.tanh(obs[:, :ad] * w + obs[:, od - ad:] * 0.5)
    oa = ea.reset_tensors()
    for k in range(21):
        a_ref = policy(oa)
        oa, ra, da = ea.step_tensors(a_ref.contiguous())
    st = ChainStepper(eb, n_chains=2)
    graph, (ob, rb, db, a_last) = st.capture_closed_loop(policy, 7, joined=joined)     # nothing runs during capture
    eb.reset_tensors()                      # ONE reset, like ea (an env's episode counter keys its Philox stream)
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(a_ref, a_last)
    assert np.array_equal(ea.get_state(), eb.get_state())
    assert int(ea.handle.episode_counter().max()) >= 3
    ea.close(); eb.close()


def test_range_arguments_are_checked():
    h = _lib.Handle(P.make_config("rov6", 1000, auto_reset=False, max_steps=10))
    a = h.dev_alloc(1000 * 6 * 4); o = h.dev_alloc(1000 * 9 * 4); r = h.dev_alloc(4000); d = h.dev_alloc(1000)
    for first, cnt in [(-64, 64), (32, 64), (960, 64), (0, 0), (0, 1001)]:
        with pytest.raises(_lib.MvrlError):
            h.step_range_dev(first, cnt, a, o, r, d, None)
    h.step_range_dev(960, 40, a, o, r, d, None)   # ragged tail range is fine
    h.synchronize()
    for p in (a, o, r, d):
        h.dev_free(p)
    h.close()


def test_host_call_after_chains_waits_for_every_stream():
    """get_state() (host-pointer entry point, the handle's own stream) after range launches on two caller streams must see
    both: the handle keeps one join event per caller stream."""
    import torch
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    n = 262144
    e = MarineVecEnv("rov6", n, seed=3, infos="lean")
    e.reset_tensors()
    acts = torch.rand((n, 6), device="cuda") * 2 - 1
    obs, rew, done = e._ensure_tensors()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ptr = (acts.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    for _ in range(20):
        e.handle.step_range_dev(0, n // 2, *ptr, s0.cuda_stream)
        e.handle.step_range_dev(n // 2, n // 2, *ptr, s1.cuda_stream)
    cnt = e.handle.step_counter()          # no explicit synchronisation
    assert (cnt == 20).all()
    e.close()


def test_delay_kernel_holds_its_stream_and_always_ends():
    """mvrl_delay_dev: one idle wave that occupies a stream for about the requested time (used to phase chains against each
    other without a cross-stream dependency); out-of-range requests are refused; work queued behind it on the same stream
    starts late, work on another stream does not wait."""
    import time
    import torch
    h = _lib.Handle(P.make_config("rov6", 64, auto_reset=False, max_steps=10))
    with pytest.raises(_lib.MvrlError):
        h.delay_dev(10001, None)
    with pytest.raises(_lib.MvrlError):
        h.delay_dev(-1, None)
    # HIP maps streams onto a few hardware queues (4 by default), round-robin over every stream the PROCESS ever created: two
    # streams that share a queue serialise.  Four candidates for "another stream": at least one sits on another queue than s0.
    s0, others = torch.cuda.Stream(), [torch.cuda.Stream() for _ in range(4)]
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    f = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in others]
    h.delay_dev(200, s0.cuda_stream)                     # warm-up (first launch of the kernel)
    torch.cuda.synchronize()
    e0.record(s0)
    for (fa, _), s in zip(f, others):
        fa.record(s)
    h.delay_dev(3000, s0.cuda_stream)
    e1.record(s0)
    for (_, fb), s in zip(f, others):
        fb.record(s)
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 1.0                # it ended
    assert 2.5 < e0.elapsed_time(e1) < 6.0               # ms: the stream was held for about 3 ms
    assert min(fa.elapsed_time(fb) for fa, fb in f) < 1.0   # another stream did not wait
    h.delay_dev(0, s0.cuda_stream)                       # no-op
    h.close()
