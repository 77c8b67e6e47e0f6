"""The N > 1 path on CPU: world_size-2 `gloo` process group, contiguous sharding, the per-step gather of
(obs, reward, done) to rank 0 / to all ranks, ragged shards.  The per-rank stepper here is the CPU oracle (tests may
use it); on the GPU box the same ShardedVecEnv wraps MarineVecEnv (tests/test_gpu_api.py, bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from marinevehiclereinforcementlearning_amd import distributed as D


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShard:
    """CPU stepper with the step_tensors/reset_tensors interface; per-env inputs are functions of the GLOBAL id."""

    has_reward = False      # 6-DoF: reward = 0. (6DoF.py:575) -> ShardedVecEnv builds the 37-byte message without a reward plane

    def __init__(self, offset, count, n_global, supports_out=False):
        from oracle import oracle as orc
        self.supports_out = supports_out
        self.env = orc.OracleRovEnv(6, count, "f64", n_substeps=2, max_steps=10 ** 9)
        rng = np.random.default_rng(5)
        init = np.concatenate([(rng.random((n_global, 6)) - 0.5) * 10, rng.random((n_global, 3)) * 2 * np.pi], axis=1)
        self.init = init[offset:offset + count]

    def reset_tensors(self):
        return torch.from_numpy(self.env.reset(self.init)).float()

    def step_tensors(self, actions, out=None):
        o, r, d = self.env.step(actions.numpy().astype(np.float64))
        res = torch.from_numpy(o).float(), torch.from_numpy(r).float(), torch.from_numpy(d)
        if out is None:
            return res
        for dst, src in zip(out, res):  # the producer writes the gather message in place (MarineVecEnv out=)
            assert dst.is_contiguous() and dst.shape == src.shape and dst.dtype == src.dtype
            dst.copy_(src)
        return out


def worker(rank, world, port, n_global, mode, q, inplace=False, scatter=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    env = D.ShardedVecEnv(lambda off, cnt, rk: OracleShard(off, cnt, n_global, inplace), n_global, 9, torch.device("cpu"), gather=mode,
                          scatter_act_dim=6 if scatter else None)
    if env.gather is not None:   # obs 36 B + done 1 B per env, no reward plane
        assert not env.gather.reward_plane and env.gather.msg_bytes == (env.gather.cmax * 37 + 15) // 16 * 16
    acts = torch.from_numpy(np.random.default_rng(9).uniform(-1, 1, size=(4, n_global, 6))).float()
    def keep(x):  # gathered outputs are views into the receive buffer, valid until the next call
        return None if x is None else (x.clone() if torch.is_tensor(x) else tuple(t.clone() for t in x))
    outs = [keep(env.reset())]
    for s in range(4):
        if scatter:   # only rank 0 knows the actions (the SB3 process); the shards arrive by dist.scatter
            outs.append(keep(env.step_global(acts[s] if rank == 0 else None)))
        else:
            outs.append(keep(env.step(acts[s, env.offset:env.offset + env.count])))
    if rank == 0 or mode == "all":
        q.put((rank, outs[0].numpy().copy(), [tuple(t.numpy().copy() for t in o) for o in outs[1:]]))
    dist.barrier()
    dist.destroy_process_group()


def run_world(n_global, mode, world=2, inplace=False, scatter=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n_global, mode, q, inplace, scatter)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world if mode == "all" else 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def single_process(n_global):
    sh = OracleShard(0, n_global, n_global)
    acts = torch.from_numpy(np.random.default_rng(9).uniform(-1, 1, size=(4, n_global, 6))).float()
    out0 = sh.reset_tensors().numpy()
    outs = [tuple(t.numpy() for t in sh.step_tensors(acts[s])) for s in range(4)]
    return out0, outs


def test_shard_range_partitions_everything():
    for n, w in [(10, 4), (8388608, 8), (7, 8), (1, 1), (1048577, 2)]:
        rs = [D.shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and sum(c for _, c in rs) == n
        assert all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        assert max(c for _, c in rs) - min(c for _, c in rs) <= 1


@pytest.mark.parametrize("n_global,mode,inplace", [(16, "root", False), (11, "root", False), (11, "all", False),
                                                  (11, "root", True), (16, "all", True)])
def test_two_rank_gather_equals_single_process(n_global, mode, inplace):
    """Concatenation of the shards == the unsharded batch, bit for bit (even and ragged shards; producer-side copy
    and the zero-copy path where the stepper writes the gather message in place)."""
    ref0, ref = single_process(n_global)
    for rank, got0, got in run_world(n_global, mode, inplace=inplace):
        assert np.array_equal(got0, ref0)
        for s in range(4):
            assert np.array_equal(got[s][0], ref[s][0])
            assert np.array_equal(got[s][1], ref[s][1])
            assert np.array_equal(got[s][2], ref[s][2].astype(np.uint8))


@pytest.mark.parametrize("n_global", [16, 11])
def test_two_rank_scatter_step_gather_equals_single_process(n_global):
    """Rank 0 alone holds the action batch: scatter -> step -> gather == the unsharded batch, bit for bit."""
    ref0, ref = single_process(n_global)
    (rank, got0, got), = run_world(n_global, "root", inplace=True, scatter=True)
    assert np.array_equal(got0, ref0)
    for s in range(4):
        for j in range(2):
            assert np.array_equal(got[s][j], ref[s][j])


@pytest.mark.parametrize("n_global,mode,scatter", [(67, "root", False), (67, "all", False), (13, "root", True)])
def test_eight_rank_gather_equals_single_process(n_global, mode, scatter):
    """World size 8 - the size of BASELINE configs[4] and half of the reference's SubprocVecEnv pool (tag/main_00_sbl.py:145, nProc 16) -
    with ragged shards (67 = 3 x 9 + 5 x 8; 13 = 5 x 2 + 3 x 1): shard ranges, padded gather message, scatter of the actions from rank 0,
    concatenation == the unsharded batch bit for bit on rank 0 (root) / on every rank (all)."""
    rs = [D.shard_range(n_global, r, 8) for r in range(8)]
    assert sum(c for _, c in rs) == n_global and max(c for _, c in rs) - min(c for _, c in rs) == 1
    ref0, ref = single_process(n_global)
    res = run_world(n_global, mode, world=8, inplace=True, scatter=scatter)
    assert len(res) == (8 if mode == "all" else 1)
    for rank, got0, got in res:
        assert np.array_equal(got0, ref0)
        for s in range(4):
            assert np.array_equal(got[s][0], ref[s][0])
            assert np.array_equal(got[s][2], ref[s][2].astype(np.uint8))


def test_scatter_single_process_passthrough():
    sc = D.ActionScatter(5, 3, torch.device("cpu"))
    a = torch.rand(5, 3)
    assert torch.equal(sc.exchange(a), a) and sc.bytes_per_step() == 5 * 3 * 4


def test_gather_single_process_passthrough():
    g = D.OutputGather(5, 3, torch.device("cpu"))
    obs, rew, done = torch.rand(5, 3), torch.rand(5), torch.tensor([0, 1, 0, 3, 0], dtype=torch.uint8)
    o, r, d = g(obs, rew, done)
    assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(d, done)
    assert g.bytes_per_step() == (5 * 3 * 4 + 5 * 4 + 5 + 15) // 16 * 16
    o2, r2, d2 = g.out_views()          # zero-copy producer views alias the message
    o2.fill_(2.0); r2.fill_(3.0); d2.fill_(1)
    g.exchange()
    o, r, d = g.unpack()
    assert float(o.min()) == 2.0 and float(r.max()) == 3.0 and int(d.sum()) == 5


def test_gather_without_reward_plane():
    """Rigid-body message: obs | done.  The producer's reward view is scratch outside the message, receivers get zeros."""
    g = D.OutputGather(5, 9, torch.device("cpu"), reward_plane=False)
    assert g.msg_bytes == (5 * 36 + 5 + 15) // 16 * 16 and g.bytes_per_step() == g.msg_bytes
    o2, r2, d2 = g.out_views()
    o2.fill_(2.0); r2.fill_(7.0); d2.fill_(3)         # a producer may write anything into its reward output
    g.exchange()
    o, r, d = g.unpack()
    assert float(o.min()) == 2.0 and float(r.abs().max()) == 0.0 and int(d.sum()) == 15
    obs, done = torch.rand(5, 9), torch.tensor([0, 1, 0, 3, 0], dtype=torch.uint8)
    o, r, d = g(obs, torch.zeros(5), done)
    assert torch.equal(o, obs) and torch.equal(d, done) and not r.any()


# ---- GatherPipeline: step k+1 overlapped with gather k on two buffers (bench.py --gpus N, with_gather) --------------------
class _SimStreams:
    """Asynchronous two-stream machine with HIP's event semantics, for testing the pipeline's ordering protocol without a
    GPU: operations are QUEUED per stream at call time and EXECUTED later by `drain_random`, which repeatedly picks a
    random stream whose head operation is allowed to run.  record(ev) bumps the event's version when it executes;
    wait(ev) captures, at enqueue time, the version of the most recent record ENQUEUED so far and blocks its stream until
    that version has executed."""

    def __init__(self, rng):
        self.rng = rng
        self.q = {"main": [], "side": []}
        self.side = "side"
        self.enq = {}      # event -> versions enqueued
        self.done = {}     # event -> versions executed

    def main(self):
        return "main"

    def new_event(self):
        ev = object()
        self.enq[ev], self.done[ev] = 0, 0
        return ev

    def record(self, ev, stream):
        self.enq[ev] += 1
        ver = self.enq[ev]
        self.q[stream].append(("record", ev, ver))

    def wait(self, stream, ev):
        self.q[stream].append(("wait", ev, self.enq[ev]))

    def run(self, stream, fn):
        self.q[stream].append(("run", fn, None))

    def join(self, stream, other):
        ev = self.new_event()
        self.record(ev, other)
        self.wait(stream, ev)

    def drain_random(self):
        while self.q["main"] or self.q["side"]:
            ready = []
            for name, ops in self.q.items():
                if ops and (ops[0][0] != "wait" or self.done[ops[0][1]] >= ops[0][2]):
                    ready.append(name)
            assert ready, "deadlock: every stream waits on an event nobody will record"
            name = ready[int(self.rng.integers(len(ready)))]
            kind, a, ver = self.q[name].pop(0)
            if kind == "run":
                a()
            elif kind == "record":
                self.done[a] = ver


class _TagGather:
    """Stands in for OutputGather: the 'message' is a one-element tensor; exchange() ships whatever it holds."""

    def __init__(self, sent):
        self.buf = torch.zeros(1)
        self.sent = sent

    def out_views(self):
        return self.buf

    def exchange(self):
        self.sent.append(float(self.buf[0]))


@pytest.mark.parametrize("seed", range(8))
def test_gather_pipeline_ordering_under_random_async_schedules(seed):
    """40 steps through the two-buffer pipeline on the simulated asynchronous backend: in every legal execution order
    message k must leave with the data of step k - i.e. producer k+2 never overwrites buffer k & 1 before gather k has
    read it, and gather k never runs before producer k has written (event re-use at k >= 2 included)."""
    rng = np.random.default_rng(seed)
    sim = _SimStreams(rng)
    sent = []
    pipe = D.GatherPipeline([_TagGather(sent), _TagGather(sent)], streams=sim)
    K = 40
    for k in range(K):
        pipe.step(lambda out, k=k: out.fill_(float(k + 1)))
        if rng.random() < 0.3:          # the host sometimes runs far ahead, sometimes the device catches up
            sim.drain_random()
    pipe.drain()
    sim.drain_random()
    assert sent == [float(k + 1) for k in range(K)]


def test_gather_pipeline_detects_a_missing_wait():
    """The simulator is sharp enough to see the race the protocol prevents: without the `buffer free` wait some schedule
    overwrites a message before it was sent."""
    class Broken(D.GatherPipeline):
        def step(self, produce):
            b = self.k & 1
            g, s = self.g[b], self.s
            s.run(s.main(), lambda: produce(g.out_views()))       # no wait on ev_gather[b]
            s.record(self.ev_step[b], s.main())
            s.wait(s.side, self.ev_step[b])
            s.run(s.side, g.exchange)
            s.record(self.ev_gather[b], s.side)
            self.k += 1
    raced = False
    for seed in range(20):
        sim = _SimStreams(np.random.default_rng(seed))
        sent = []
        pipe = Broken([_TagGather(sent), _TagGather(sent)], streams=sim)
        for k in range(12):
            pipe.step(lambda out, k=k: out.fill_(float(k + 1)))
        pipe.drain()
        sim.drain_random()
        raced |= sent != [float(k + 1) for k in range(12)]
    assert raced


def test_gather_pipeline_program_order_on_cpu_tensors():
    g = [D.OutputGather(6, 3, torch.device("cpu")) for _ in range(2)]
    pipe = D.GatherPipeline(g)
    for k in range(5):
        def produce(out, k=k):
            o, r, d = out
            o.fill_(float(k)); r.fill_(float(10 + k)); d.fill_(k % 2)
        assert pipe.step(produce) == k & 1
        o, r, d = pipe.latest().unpack()
        assert float(o.min()) == k and float(r.max()) == 10 + k and int(d[0]) == k % 2
        if k:                            # the other buffer still holds the previous step's message
            o_prev, _, _ = g[(k - 1) & 1].unpack()
            assert float(o_prev.max()) == k - 1
    pipe.drain()
