"""Size-independent properties at BASELINE.json's full sizes (262 144 / 1 048 576 envs) on the GPU:
replication invariance (identical lanes stay bit-identical wherever they sit in the batch), agreement of the
three kernel flavours, finiteness/boundedness, episode accounting."""
import numpy as np
import pytest

from marinevehiclereinforcementlearning_amd import _lib, params as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,n", [("rov6", 262144), ("rov3", 65536), ("rov6", 1048576)])
def test_replicated_lanes_stay_identical(model, n):
    """Tile 64 distinct (init, action-sequence) pairs across the whole batch: every replica must match its
    prototype bit for bit after 10 steps - exercises every block/wave position of the full-size launch."""
    dof = 6 if model == "rov6" else 3
    idim = 9 if dof == 6 else 5
    rng = np.random.default_rng(4)
    proto_init = np.concatenate([(rng.random((64, idim - (dof // 2 if dof == 6 else 1))) - 0.5) * 10,
                                 rng.random((64, 3 if dof == 6 else 1)) * 2 * np.pi], axis=1).astype(np.float32)
    proto_act = rng.uniform(-1, 1, size=(10, 64, dof)).astype(np.float32)
    reps = n // 64
    h = _lib.Handle(P.make_config(model, n, auto_reset=False, max_steps=10 ** 9, use_flow=False))
    h.reset(init=np.tile(proto_init, (reps, 1)))
    for s in range(10):
        obs, _, _ = h.step(np.tile(proto_act[s], (reps, 1)))
    st = h.get_state().reshape(-1, reps, 64)
    assert np.isfinite(st).all()
    assert np.array_equal(st, np.broadcast_to(st[:, :1, :], st.shape))
    o = obs.reshape(reps, 64, -1)
    assert np.array_equal(o, np.broadcast_to(o[:1], o.shape))
    assert np.all(np.abs(obs) <= 1.0)
    h.close()


def test_kernel_flavours_agree_at_full_size():
    """baked (literal constants) vs ctrl (literal vehicle, run-time controller) vs sym (run-time constants): the same numbers
    up to one fp32 ulp in one constant, the same arithmetic - so identical to fp32 rounding."""
    n = 262144
    rng = np.random.default_rng(9)
    init = np.concatenate([(rng.random((n, 6)) - 0.5) * 10, rng.random((n, 3)) * 2 * np.pi], axis=1).astype(np.float32)
    acts = rng.uniform(-1, 1, size=(5, n, 6)).astype(np.float32)
    hb = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False))
    pc = P.rov6_params()
    pc.kp[0] = 25.0 * (1 + 1e-7)     # one fp32 ulp: no longer the reference's controller -> not "baked", vehicle untouched
    hc = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=pc))
    ps = P.rov6_params()
    ps.dlin[0] = ps.dlin[0] * (1 + 1.5e-7)   # one ulp in a vehicle constant (linear surge damping) -> structured run-time constants
    hs = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=ps))
    assert "baked" in hb.variant and "ctrl" in hc.variant and "sym" in hs.variant, (hb.variant, hc.variant, hs.variant)
    # ... and the same one-ulp vehicle compiled at run time with its constants as literals (mvrl_specialize)
    hj = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=False, rov6=ps))
    assert "jit-sym" in hj.specialize()
    for h in (hb, hc, hs, hj):
        h.reset(init=init)
    for s in range(5):
        for h in (hb, hc, hs, hj):
            h.step(acts[s])
    a = hb.get_state()[:12]
    hj_state = hj.get_state()[:12]
    hj.close()
    # literal constants either way, one constant apart by one ulp: the specialised kernel tracks the baked one closely
    assert np.median(np.abs(hj_state - a).max(axis=0)) < 2e-6
    for h in (hc, hs):
        b = h.get_state()[:12]
        d = np.abs(a - b)
        d[3:6] = np.minimum(d[3:6], np.abs(d[3:6] - 2 * np.pi))
        bad = (d / np.maximum(1, np.abs(a))).max(axis=0) > 1e-4
        assert bad.mean() < 1e-3, (h.variant, bad.sum())
    for h in (hb, hc, hs):
        h.close()


def test_episode_accounting_at_full_size():
    n, max_steps = 1048576, 7
    h = _lib.Handle(P.make_config("rov6", n, max_steps=max_steps, auto_reset=True, seed=5, use_flow=False))
    h.reset()
    a = np.random.default_rng(0).uniform(-1, 1, size=(n, 6)).astype(np.float32)
    total_done = 0
    for s in range(15):
        _, r, d = h.step(a)
        total_done += int((d != 0).sum())
        assert not r.any()
    assert total_done == 2 * n
    ist = h.get_state()[-1].view(np.int32)
    assert np.all(ist == 15 % max_steps)
    h.close()


def test_auvenv_half_turn_symmetry_at_full_size():
    """A symmetry the equations have and no implementation detail should break (the reference probes the observation
    side of it in tag/script_5_testTransformations.py): without current, turning the whole scene by pi about the target
    - (x, y, psi, psi_target) -> (-x, -y, psi + pi, psi_target + pi), global force actions (a0, a1, a2) -> (-a0, -a1, a2) -
    maps a trajectory onto its mirror image: observations 0, 1, 4..7 change sign, 2, 3, 8 and the reward stay.
    1 048 576 envs x 40 steps; equality up to the fp32 rounding of psi + pi."""
    n, steps = 1048576, 40
    rng = np.random.default_rng(12)
    auv = P.auv_params(noiseMagCoeffs=0.1, noiseMagActuation=0.1, stopOnBoundsExceeded=False)
    init = np.zeros((n, 16), np.float32)
    init[:, :2] = (rng.random((n, 2)) - 0.5) * 0.5
    init[:, 2] = rng.random(n) * np.pi              # psi in [0, pi): psi + pi stays inside [0, 2 pi)
    init[:, 3] = init[:, 2] + (rng.random(n) - 0.5) * 2.0     # heading error within +-1 rad: away from the +-pi tie
    init[:, 5:] = 1.0 + 0.05 - rng.random((n, 11)) * 0.1
    mirror = init.copy()
    mirror[:, :2] *= -1
    mirror[:, 2] += np.float32(np.pi)
    mirror[:, 3] += np.float32(np.pi)
    sign = np.array([-1, -1, 1, 1, -1, -1, -1, -1, 1, 1, 1], np.float32)
    ha = _lib.Handle(P.make_config("auv", n, dt=0.02, auto_reset=False, max_steps=10 ** 9, use_flow=False, auv=auv))
    hb = _lib.Handle(P.make_config("auv", n, dt=0.02, auto_reset=False, max_steps=10 ** 9, use_flow=False, auv=auv))
    oa, ob = ha.reset(init=init), hb.reset(init=mirror)
    assert np.max(np.abs(oa - ob * sign)) < 2e-5
    worst_o = worst_r = 0.0
    tie = np.zeros(n, bool)     # lanes whose heading error has come within 1e-3 of +-pi: angleError's branch cut
    for k in range(steps):
        a = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        b = a * np.array([-1, -1, 1], np.float32)
        oa, ra, _ = ha.step(a)
        ob, rb, _ = hb.step(b)
        herr = ha.get_state()[P.STATE_PLANES[P.MODEL_AUV]["herr_o"]]
        tie |= np.abs(herr) > np.pi - 1e-3
        ok = ~tie
        worst_o = max(worst_o, float(np.max(np.abs(oa[ok] - (ob * sign)[ok]))))
        worst_r = max(worst_r, float(np.max(np.abs(ra[ok] - rb[ok]) / np.maximum(1.0, np.abs(ra[ok])))))
    # the heading-error-change observation is scaled by 1/(2 deg): 28.6 x the 4.8e-7 resolution of an angle near 2 pi
    assert tie.mean() < 1e-3, tie.sum()
    assert worst_o < 1e-4, worst_o
    assert worst_r < 1e-4, worst_r
    ha.close(); hb.close()


# ---- the instances bench.py times, at the size it times them, against the oracle -------------------------------------------
def _bench_flow():
    from marinevehiclereinforcementlearning_amd.flow import ReconstructedFlow
    flow = ReconstructedFlow.synthetic(n_modes=8, n_time=2000, device=0)     # exactly bench.py's table
    flow.scale(11., 1., 2., translate=(-1.65, -1.1))
    return flow


def test_benched_c4_instance_at_full_size_vs_oracle(oracle_mod):
    """BASELINE configs[3] exactly as bench.py builds and steps it - MarineVecEnv("rov6", 1 048 576 envs, seed 12345,
    n_substeps 4, FAITHFUL, fp32, baked constants, 2000-snapshot turbulence table, random Philox initial paths / attitudes /
    time offsets, ring of uniform(-1,1) action batches from mvrl_fill_uniform_dev, two chains of lane ranges) - for 10 steps.
    A fixed random subsample of 4096 lanes (whole batch positions: first / last wave, chain boundary included) is re-run by
    the fp64 oracle from the same initial planes, actions and time offsets; tolerance 1e-5 with the discontinuity audit."""
    import torch
    from marinevehiclereinforcementlearning_amd.chains import ChainStepper
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    from .parity_util import OutlierAudit
    n, steps, seed = 1048576, 10, 12345
    flow = _bench_flow()
    env = MarineVecEnv("rov6", n, seed=seed, n_substeps=4, control_mode="faithful", flow=flow, device=0, infos="lean")
    assert env.variant == "rov6/baked/faithful+flow"
    h = env.handle
    stream = torch.cuda.current_stream().cuda_stream
    ring = torch.empty((8, n, 6), dtype=torch.float32, device="cuda")
    for r in range(8):
        h.fill_uniform_dev(ring[r].data_ptr(), n * 6, seed, r, -1.0, 1.0, stream)
    env.reset_tensors()
    rng = np.random.default_rng(3)
    lanes = np.unique(np.concatenate([np.arange(64), np.arange(n - 64, n), np.arange(n // 2 - 64, n // 2 + 64),
                                      rng.choice(n, 4096 - 256, replace=False)]))
    st0 = env.get_state()
    pl = P.STATE_PLANES[P.MODEL_ROV6]
    init = np.concatenate([st0[pl["path"]][:, lanes].T, st0[pl["setpoint"]][3:, lanes].T], axis=1).astype(np.float64)
    toff = st0[pl["toffset"]][lanes].astype(np.float64)
    assert np.all(st0[pl["y"]] == 0) and toff.min() >= 0 and toff.max() > 1.0       # random offsets within flow.time[nT // 4]
    ref = oracle_mod.OracleRovEnv(6, len(lanes), "f64", n_substeps=4, max_steps=250,
                                  flow=oracle_mod.FlowTable(flow.table_uv().astype(np.float64), flow.dt, flow.dx, flow.dy))
    ref_obs0 = ref.reset(init, toffset=toff)
    obs_t = env._ensure_tensors()[0]
    assert np.max(np.abs(obs_t[torch.as_tensor(lanes, device="cuda")].cpu().numpy() - ref_obs0)) < 1e-5
    acts = ring[:, torch.as_tensor(lanes, device="cuda")].cpu().numpy().astype(np.float64)
    stepper = ChainStepper(env, n_chains=2)
    audit = OutlierAudit(len(lanes), 1e-5, dof=6)
    for k in range(steps):
        stepper.fork()
        obs, rew, done = stepper.step(ring[k % 8])
        stepper.join()
        ref_obs, _, ref_done = ref.step(acts[k % 8])
        y = env.get_state()[:12][:, lanes].T.astype(np.float64)
        d = np.abs(y - ref.y)
        d[:, 3:6] = np.minimum(d[:, 3:6], np.abs(d[:, 3:6] - 2 * np.pi))
        audit.update((d / np.maximum(1.0, np.abs(ref.y))).max(axis=1), ref.margins)
        on = ~audit.bad
        o = obs[torch.as_tensor(lanes, device="cuda")].cpu().numpy()
        assert np.max(np.abs(o - ref_obs)[on]) < 2e-5, k
        assert not done.any().item() and not ref_done.any()
    print("C4 at full size: " + audit.report())
    from .parity_util import make_resolver
    audit.assert_explained(max_share=0.004, max_smooth_share=0.002,
                           resolver=make_resolver(oracle_mod, 6, init, [acts[k % 8] for k in range(steps)], dict(n_substeps=4, flow=ref.flow),
                                                  toffset=toff))
    assert torch.isfinite(obs).all().item()
    env.close()


def test_benched_c4_instance_stays_finite_over_whole_episodes():
    """BASELINE configs[3] as bench.py steps it, for 520 steps (two whole 250-step episodes with their auto-resets): EVERY plane of
    EVERY env stays finite and the speeds stay of the order of the current.  With the reference's linear extrapolation outside the
    3.3 m x 2.2 m / 44 s table this workload ended every episode non-finite (11 % of the envs after 100 steps, all after 224,
    measured with the fp64 oracle) - invisible in the outputs because observations are clipped to +-1; the composition holds /
    reflects outside the table instead."""
    import torch
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    n, seed = 1048576, 12345
    env = MarineVecEnv("rov6", n, seed=seed, n_substeps=4, control_mode="faithful", flow=_bench_flow(), device=0, infos="lean")
    h = env.handle
    stream = torch.cuda.current_stream().cuda_stream
    ring = torch.empty((8, n, 6), dtype=torch.float32, device="cuda")
    for r in range(8):
        h.fill_uniform_dev(ring[r].data_ptr(), n * 6, seed, r, -1.0, 1.0, stream)
    env.reset_tensors()
    worst_speed, far = 0.0, 0.0
    for k in range(520):
        obs, rew, done = env.step_tensors(ring[k % 8])
        if k % 40 == 39 or k in (248, 249, 250, 499, 519):
            st = torch.from_numpy(env.get_state()[:36])
            assert torch.isfinite(st).all().item(), k
            worst_speed = max(worst_speed, float(st[6:9].abs().max()))
            far = max(far, float(st[0].abs().max()))
            assert bool(done.all().item()) == (k in (249, 499)), k       # all envs end their episodes together
    print(f"520 steps of C4: largest body speed {worst_speed:.2f} m/s, farthest x {far:.1f} m")
    assert worst_speed < 5.0 and far > 10.0            # bounded current; and yes, the vehicles are carried far outside the table
    env.close()


def test_replicated_lanes_with_turbulence_at_full_size():
    """The replication property on the flow-enabled baked instance (the one bench.py times): 64 prototypes (initial paths,
    time offsets, action sequences) tiled over 1 048 576 lanes stay bit-identical wherever they sit."""
    n, reps = 1048576, 1048576 // 64
    flow = _bench_flow()
    rng = np.random.default_rng(8)
    proto_init = np.concatenate([(rng.random((64, 6)) - 0.5) * 2.0, rng.random((64, 3)) * 2 * np.pi], axis=1).astype(np.float32)
    proto_toff = (rng.random(64) * 10.0).astype(np.float32)
    proto_act = rng.uniform(-1, 1, size=(10, 64, 6)).astype(np.float32)
    h = _lib.Handle(P.make_config("rov6", n, auto_reset=False, max_steps=10 ** 9, use_flow=True))
    h.set_flow(flow.table_uv(), flow.dt, flow.dx, flow.dy)
    assert h.variant == "rov6/baked/faithful+flow"
    h.reset(init=np.tile(proto_init, (reps, 1)))
    st = h.get_state()
    st[P.STATE_PLANES[P.MODEL_ROV6]["toffset"]] = np.tile(proto_toff, reps)
    h.set_state(st)
    for s in range(10):
        obs, _, _ = h.step(np.tile(proto_act[s], (reps, 1)))
    st = h.get_state().reshape(-1, reps, 64)
    assert np.isfinite(st).all() and np.abs(st[6:12]).max() > 1e-3        # the vehicles moved
    assert np.array_equal(st, np.broadcast_to(st[:, :1, :], st.shape))
    o = obs.reshape(reps, 64, -1)
    assert np.array_equal(o, np.broadcast_to(o[:1], o.shape))
    h.close()


def test_benched_auv_instance_at_full_size_vs_oracle(oracle_mod):
    """AuvEnv + turbulence as `bench.py --workload auv` builds it (1 048 576 envs, random Philox resets with the reference's
    noise magnitudes switched off as in the bench, random actions): 4096 sampled lanes re-run by the fp64 oracle from the
    same initial planes for 10 steps - poses, observations, rewards, done flags."""
    import torch
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    n, steps, seed = 1048576, 10, 12345
    flow = _bench_flow()
    env = MarineVecEnv("auv", n, seed=seed, flow=flow, device=0, infos="lean")
    h = env.handle
    ring = torch.empty((8, n, 3), dtype=torch.float32, device="cuda")
    for r in range(8):
        h.fill_uniform_dev(ring[r].data_ptr(), n * 3, seed, r, -1.0, 1.0, torch.cuda.current_stream().cuda_stream)
    obs0 = env.reset_tensors().clone()
    rng = np.random.default_rng(4)
    lanes = np.unique(np.concatenate([np.arange(64), np.arange(n - 64, n), rng.choice(n, 4096 - 128, replace=False)]))
    li = torch.as_tensor(lanes, device="cuda")
    st0 = env.get_state()
    pl = P.STATE_PLANES[P.MODEL_AUV]
    init = np.zeros((len(lanes), 16))
    init[:, :3] = st0[pl["pose"]][:3, lanes].T
    init[:, 3] = st0[pl["heading_target"]][lanes]
    init[:, 4] = st0[pl["toffset"]][lanes]
    init[:, 5:] = st0[pl["mult"]][:, lanes].T
    ref = oracle_mod.OracleAuvEnv(len(lanes), "f64", dt=env.dt, max_steps=250,
                                  flow=oracle_mod.FlowTable(flow.table_uv().astype(np.float64), flow.dt, flow.dx, flow.dy))
    assert np.max(np.abs(obs0[li].cpu().numpy() - ref.reset(init))) < 1e-5
    acts = ring[:, li].cpu().numpy().astype(np.float64)
    alive = np.ones(len(lanes), bool)
    for k in range(steps):
        obs, rew, done = env.step_tensors(ring[k % 8])
        o_ref, r_ref, d_ref = ref.step(acts[k % 8])
        d_gpu = done[li].cpu().numpy()
        alive &= ((d_gpu != 0) == (d_ref != 0)) & (d_ref == 0)      # lanes that finished were auto-reset on the GPU: drop them
        a = alive
        pose = env.get_state()[:6][:, lanes].T.astype(np.float64)
        dd = np.abs(pose - ref.pose)
        dd[:, 2] = np.minimum(dd[:, 2], np.abs(dd[:, 2] - 2 * np.pi))
        assert (dd / np.maximum(1.0, np.abs(ref.pose)))[a].max() < 1e-5, k
        assert np.max(np.abs(obs[li].cpu().numpy() - o_ref)[a]) < 3e-5, k
        assert np.max((np.abs(rew[li].cpu().numpy() - r_ref) / np.maximum(1.0, np.abs(r_ref)))[a]) < 3e-5, k
    assert alive.mean() > 0.99
    env.close()


def test_chains_soak_bit_identical_at_full_size():
    """Race check at BASELINE's full size: 1 048 576 6-DoF envs with turbulence stepped 600 times - three generations of
    random auto-resets - once as whole-batch launches on one stream and once as two independent chains of lane ranges that
    run ahead of each other on two streams (no host synchronisation in between).  Final state planes, episode counters,
    terminal observations and the last outputs must agree bit for bit."""
    import torch
    from marinevehiclereinforcementlearning_amd.chains import ChainStepper
    from marinevehiclereinforcementlearning_amd.vec_env import MarineVecEnv
    n, steps = 1048576, 600
    flow = _bench_flow()
    kw = dict(seed=77, maxSteps=200, flow=flow, infos="lean")
    a, b = MarineVecEnv("rov6", n, **kw), MarineVecEnv("rov6", n, **kw)
    ring = torch.empty((8, n, 6), dtype=torch.float32, device="cuda")
    for r in range(8):
        a.handle.fill_uniform_dev(ring[r].data_ptr(), n * 6, 5, r, -1.0, 1.0, torch.cuda.current_stream().cuda_stream)
    a.reset_tensors(); b.reset_tensors()
    torch.cuda.synchronize()
    for k in range(steps):
        oa, ra, da = a.step_tensors(ring[k % 8])
    st = ChainStepper(b, n_chains=2)
    st.fork()
    st.phase_delay(60.0)
    for k in range(steps):
        ob, rb, db = st.step(ring[k % 8])
    st.join()
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa, sb)
    assert np.array_equal(a.handle.terminal_obs(), b.handle.terminal_obs())
    ep = a.handle.episode_counter()
    assert int(ep.min()) == 4 and int(ep.max()) == 4          # the initial reset + three auto-resets, every env
    assert torch.isfinite(oa).all().item()
    a.close(); b.close()
