"""Sharding of one environment batch over the GPUs of a node: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" on CPU for tests), contiguous shards, and ONE collective per env step - the gather
of (observation, reward, done) rows to rank 0 over xGMI, which replaces the pipe recv + np.stack of SB3's
SubprocVecEnv (main_00_sbl.py:145).

The environments are independent, so stepping needs no communication; RNG streams are keyed by the GLOBAL env
index (mvrl_config.env_offset), which makes N envs on one GPU identical to the concatenation of the shards.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_global, rank, world_size):
    """Contiguous block partition; the first (n_global % world_size) ranks own one extra env."""
    base, extra = divmod(int(n_global), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def message_layout(n_global, world_size, obs_dim, reward_plane=True):
    """The gather message of one rank and step (OutputGather; the C side's mvrl_group_message_layout is the same arithmetic):
    obs[cmax, obs_dim] f32 | reward[cmax] f32 (if reward_plane) | done[cmax] u8, padded to 16 B; cmax = the largest shard."""
    cmax = max(shard_range(n_global, r, world_size)[1] for r in range(world_size))
    off_rew = cmax * int(obs_dim) * 4
    off_done = off_rew + (cmax * 4 if reward_plane else 0)
    return dict(cmax=cmax, off_reward=off_rew, off_done=off_done, msg_bytes=(off_done + cmax + 15) // 16 * 16)


def init_from_env(backend=None):
    """Join the process group torchrun set up (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


class OutputGather:
    """Per-step gather of the shards' (obs, reward, done) to `root`: ONE message per rank and step.

    The message is planar - obs[cmax, obs_dim] f32 | reward[cmax] f32 | done[cmax] u8 (padded to 16 B) - so that the
    step kernel can write its outputs straight into it (`out_views()` handed to `MarineVecEnv.step_tensors(out=...)`):
    there is no pack kernel and no extra HBM pass between the env step and the wire.  Every rank sends the same
    number of bytes (a requirement of dist.gather), sized for the largest shard.  `mode="all"` all-gathers instead
    (every rank ends up with the full batch - for a replicated policy)."""

    def __init__(self, n_global, obs_dim, device, root=0, mode="root", group=None, reward_plane=True):
        """reward_plane=False: the rigid-body environments' reward is identically 0 (6DoF.py:575, 3DoF.py:495), so their message
        carries no reward plane - obs | done, 37 B instead of 41 B per 6-DoF env on the link that bounds the gather; the
        producer's reward output goes to a scratch tensor and the receiver hands out zeros."""
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.root, self.mode, self.group = root, mode, group
        self.n_global, self.obs_dim = int(n_global), int(obs_dim)
        self.ranges = [shard_range(n_global, r, self.world) for r in range(self.world)]
        self.count = self.ranges[self.rank][1]
        self.device = device
        self.reward_plane = bool(reward_plane)
        lay = message_layout(n_global, self.world, obs_dim, self.reward_plane)
        self.cmax, self.off_rew, self.off_done, self.msg_bytes = lay["cmax"], lay["off_reward"], lay["off_done"], lay["msg_bytes"]
        self.send = torch.zeros((self.msg_bytes,), dtype=torch.uint8, device=device)
        # without a reward plane: where the step kernel's (all-zero) reward output lands, and what receivers see
        self._rew_scratch = None if self.reward_plane else torch.zeros((self.cmax,), dtype=torch.float32, device=device)
        self._rew_zero = None if self.reward_plane else torch.zeros((self.cmax,), dtype=torch.float32, device=device)
        need_recv = (mode == "all") or (self.rank == root)
        self.recv = torch.zeros((self.world, self.msg_bytes), dtype=torch.uint8, device=device) if need_recv else None

    def bytes_per_step(self):
        """Bytes arriving at the root (or at every rank in mode "all") per env step of the global batch."""
        return self.world * self.msg_bytes

    def _views(self, buf, c, sending=False):
        obs = buf[: self.off_rew].view(torch.float32).view(self.cmax, self.obs_dim)[:c]
        if self.reward_plane:
            rew = buf[self.off_rew: self.off_done].view(torch.float32)[:c]
        else:
            rew = (self._rew_scratch if sending else self._rew_zero)[:c]
        done = buf[self.off_done: self.off_done + self.cmax][:c]
        return obs, rew, done

    def out_views(self):
        """(obs[c, obs_dim] f32, reward[c] f32, done[c] u8): contiguous views INTO the send message for this rank's
        shard.  A producer that writes them (the step kernel) makes `pack` unnecessary."""
        return self._views(self.send, self.count, sending=True)

    def pack(self, obs, reward, done):
        """Copy path for producers that own their output buffers."""
        o, r, d = self._views(self.send, obs.shape[0], sending=True)
        o.copy_(obs)
        if self.reward_plane:
            r.copy_(reward)
        d.copy_(done)
        return self.send

    def exchange(self):
        if not dist.is_initialized():        # plain single process: no process group to talk to
            if self.recv is not None:
                self.recv[0].copy_(self.send)
            return
        if self.send.is_cuda and dist.get_backend(self.group) == "gloo":
            return self._exchange_via_host()  # rehearsal path: gloo has no device-side gather
        if self.mode == "all":
            dist.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.group)
        else:
            if self.rank == self.root:
                dist.gather(self.send, gather_list=list(self.recv.unbind(0)), dst=self.root, group=self.group)
            else:
                dist.gather(self.send, gather_list=None, dst=self.root, group=self.group)

    def _exchange_via_host(self):
        send = self.send.cpu()
        if self.mode == "all":
            out = torch.empty((self.world, self.msg_bytes), dtype=send.dtype)
            dist.all_gather_into_tensor(out.view(-1), send, group=self.group)
            self.recv.copy_(out)
        elif self.rank == self.root:
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.gather(send, gather_list=parts, dst=self.root, group=self.group)
            self.recv.copy_(torch.stack(parts))
        else:
            dist.gather(send, gather_list=None, dst=self.root, group=self.group)

    def unpack_shards(self):
        """On root (or everywhere in mode "all"): one (obs, reward, done) triple of VIEWS into the receive buffer per
        rank, in global env order - no copy; valid until the next exchange."""
        if self.recv is None:
            return None
        return [self._views(self.recv[r], c) for r, (_, c) in enumerate(self.ranges)]

    def unpack(self):
        """On root (or everywhere in mode "all"): (obs[N, obs_dim], reward[N], done[N] u8) in global env order
        (one concatenation of the shard views)."""
        shards = self.unpack_shards()
        if shards is None:
            return None
        if len(shards) == 1:
            return shards[0]
        return tuple(torch.cat([sh[k] for sh in shards], dim=0) for k in range(3))

    def __call__(self, obs, reward, done):
        self.pack(obs, reward, done)
        self.exchange()
        return self.unpack()


class TorchStreams:
    """The stream / event operations GatherPipeline needs, on HIP streams through torch."""

    def __init__(self, side):
        self.side = side

    def main(self):
        return torch.cuda.current_stream()

    def new_event(self):
        return torch.cuda.Event()

    def record(self, ev, stream):
        ev.record(stream)

    def wait(self, stream, ev):
        stream.wait_event(ev)          # refers to the event's most recent record at THIS moment (HIP semantics)

    def run(self, stream, fn):
        with torch.cuda.stream(stream):
            fn()                       # fn enqueues its kernels / collectives on `stream`

    def join(self, stream, other):
        stream.wait_stream(other)


class GatherPipeline:
    """Step k+1 overlapped with the gather of step k: two message buffers, the exchange on a side stream.

        pipe = GatherPipeline([OutputGather(...), OutputGather(...)], side_stream)
        for k in range(K):
            pipe.step(lambda out: env.step_tensors(actions[k], out=out))   # kernel writes message k & 1 in place
        pipe.drain()

    Ordering (events, no host synchronisation): the producer of step k may only overwrite buffer k & 1 once the exchange
    of step k-2 - the previous user of that buffer - has finished (`ev_gather[b]`); the exchange of step k starts once
    the producer has finished (`ev_step[b]`).  Both events of a buffer are re-recorded every second step; a wait refers
    to the most recent record at the moment it is enqueued, which is the one it needs.  `streams` is the stream backend:
    TorchStreams(side) on the GPU; tests drive the same protocol through a simulated asynchronous backend that executes
    the queued operations in random legal orders and checks that no buffer is overwritten before it was sent
    (tests/test_distributed_cpu.py).  Without a backend everything runs in program order (CPU tensors)."""

    def __init__(self, gathers, side_stream=None, streams=None):
        assert len(gathers) == 2
        self.g, self.k = gathers, 0
        self.s = streams if streams is not None else (TorchStreams(side_stream) if side_stream is not None else None)
        if self.s is not None:
            self.ev_step = [self.s.new_event() for _ in range(2)]
            self.ev_gather = [self.s.new_event() for _ in range(2)]

    def step(self, produce):
        """produce(out_views) enqueues the env step that writes this step's message.  Returns the buffer index used."""
        b = self.k & 1
        g = self.g[b]
        if self.s is None:
            produce(g.out_views())
            g.exchange()
        else:
            s, main, side = self.s, self.s.main(), self.s.side
            if self.k >= 2:
                s.wait(main, self.ev_gather[b])            # buffer b is free once gather k-2 has finished
            s.run(main, lambda: produce(g.out_views()))
            s.record(self.ev_step[b], main)
            s.wait(side, self.ev_step[b])
            s.run(side, g.exchange)
            s.record(self.ev_gather[b], side)
        self.k += 1
        return b

    def latest(self):
        """The OutputGather holding the most recent step's message (valid after drain(), or on the side stream)."""
        return self.g[(self.k - 1) & 1]

    def drain(self):
        if self.s is not None:
            self.s.join(self.s.main(), self.s.side)


def rccl_info(backend, local_rank):
    """What a SCALE record needs to be audited: the backend that actually ran, how many ranks took part and whether they sat
    on distinct devices (gathered over the group: each rank reports its device's PCI bus id)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    ident = None
    try:
        props = torch.cuda.get_device_properties(local_rank)
        ident = f"{getattr(props, 'pci_bus_id', '')}:{getattr(props, 'pci_device_id', '')}:{getattr(props, 'uuid', local_rank)}"
    except Exception:  # noqa: BLE001
        ident = f"cuda:{local_rank}"
    idents = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(idents, ident)
    else:
        idents = [ident]
    return {"backend": (dist.get_backend() if dist.is_initialized() else None) or backend, "world_size_seen": world,
            "ranks_on_distinct_devices": len(set(idents)) == world, "devices": idents,
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") and torch.cuda.is_available() else None}


class ActionScatter:
    """The mirror of OutputGather: a consumer that lives on one rank - SB3's single process - holds the
    action batch for ALL envs; every step rank `root` scatters the shards, one message per rank, padded to the largest
    shard.  Replaces the pipe send of SubprocVecEnv.step_async (main_00_sbl.py:145).  `local()` is this rank's
    contiguous [count, act_dim] view, valid until the next exchange."""

    def __init__(self, n_global, act_dim, device, root=0, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.root, self.group = root, group
        self.n_global, self.act_dim = int(n_global), int(act_dim)
        self.ranges = [shard_range(n_global, r, self.world) for r in range(self.world)]
        self.cmax = max(c for _, c in self.ranges)
        self.offset, self.count = self.ranges[self.rank]
        self.recv = torch.zeros((self.cmax, self.act_dim), dtype=torch.float32, device=device)
        self.send = (torch.zeros((self.world, self.cmax, self.act_dim), dtype=torch.float32, device=device)
                     if self.rank == root else None)

    def bytes_per_step(self):
        return self.world * self.cmax * self.act_dim * 4

    def exchange(self, actions_global=None):
        """root passes the [n_global, act_dim] batch (global env order); the other ranks pass nothing."""
        if self.rank == self.root:
            a = actions_global.reshape(self.n_global, self.act_dim)
            for r, (o, c) in enumerate(self.ranges):
                self.send[r, :c].copy_(a[o:o + c])
        if not dist.is_initialized():
            self.recv.copy_(self.send[0])
            return self.local()
        if self.recv.is_cuda and dist.get_backend(self.group) == "gloo":   # rehearsal path, as in OutputGather
            buf = torch.empty(self.recv.shape, dtype=torch.float32)
            dist.scatter(buf, scatter_list=list(self.send.cpu().unbind(0)) if self.rank == self.root else None,
                         src=self.root, group=self.group)
            self.recv.copy_(buf)
            return self.local()
        dist.scatter(self.recv, scatter_list=list(self.send.unbind(0)) if self.rank == self.root else None,
                     src=self.root, group=self.group)
        return self.local()

    def local(self):
        return self.recv[: self.count]


class ShardedVecEnv:
    """A global batch of `n_global` envs split over the ranks of the process group.

    `make_shard(offset, count, rank)` builds this rank's stepper: any object with `reset_tensors()` and
    `step_tensors(actions) -> (obs, reward, done)` returning torch tensors on `device` (MarineVecEnv on a GPU;
    tests inject a CPU stepper).  `step(actions_local)` steps the local shard and gathers the outputs."""

    def __init__(self, make_shard, n_global, obs_dim, device, gather="root", group=None, scatter_act_dim=None, reward_plane=None):
        """reward_plane: whether the gather message carries rewards; None = ask the shard (`has_reward`, False for the rigid-body
        models whose reward is identically 0), default True."""
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.offset, self.count = shard_range(n_global, self.rank, self.world)
        self.local = make_shard(self.offset, self.count, self.rank)
        if reward_plane is None:
            reward_plane = bool(getattr(self.local, "has_reward", True))
        self.gather = None if gather in (None, "none") else OutputGather(n_global, obs_dim, device, mode=gather, group=group,
                                                                         reward_plane=reward_plane)
        # scatter_act_dim = act_dim: step_global() takes the FULL action batch on rank 0 and scatters it (C2)
        self.scatter = None if scatter_act_dim is None else ActionScatter(n_global, scatter_act_dim, device, group=group)

    def reset(self):
        obs = self.local.reset_tensors()
        if self.gather is None:
            return obs
        z = torch.zeros(obs.shape[0], dtype=torch.float32, device=obs.device)
        out = self.gather(obs, z, z)
        return None if out is None else out[0]

    def step_global(self, actions_global=None):
        """The single-consumer loop (SB3 on rank 0): rank 0 passes actions for ALL envs, the other ranks None; shards are
        scattered, stepped, and the outputs gathered back - two collectives per step."""
        return self.step(self.scatter.exchange(actions_global))

    def step(self, actions_local):
        if self.gather is None:
            return self.local.step_tensors(actions_local)
        if getattr(self.local, "supports_out", False):
            # zero-copy: the step kernel writes obs / reward / done straight into the gather message
            self.local.step_tensors(actions_local, out=self.gather.out_views())
            self.gather.exchange()
            return self.gather.unpack()
        return self.gather(*self.local.step_tensors(actions_local))
