"""Entity re-balancing across the GPUs of a node for skewed partitions.

The reference shards by partition only (worker w trains partitions[w::W], random_effect_driver.py:60-68): with a
Zipf-distributed entity size one partition — hence one worker — can hold most of the work. Entities are
independent, so any entity may be solved on any rank; only its result has to come back to the rank that owns its
partition, which writes the model file. One round per partition step, all ranks in lockstep:

  1. all_gather of every rank's total cost (cost of an entity = its non-zeros: what the solver streams);
  2. every rank derives the same transfer matrix from the gathered loads (deterministic greedy: largest surplus
     to largest deficit) and picks which of its own entities travel — smallest first, so the giants, which no
     rank could absorb, stay where they are;
  3. the travelling entities' arrays are gathered ON THE DEVICE out of the partition's 32-bit wire form
     (gdmix_re_wire_batch: samples per entity, non-zeros per sample, int32 feature ids, values, labels, offsets,
     weights — what was uploaded for the solve anyway) and exchanged device to device: one all_to_all_single per
     array over RCCL (xGMI is point to point, 7 links per GPU: a direct all-to-all, not a ring); entity ids and
     sample ids stay at home — the solving rank does not need them, results come back by position;
  4. every rank concatenates kept + received wire arrays in HBM, widens and packs them (gdmix_re_widen,
     gdmix_re_pack) and solves them as one batch;
  5. one exchange back, device to device: per entity the coefficient count and the solver statistics, per
     coefficient the value (and variance), per feature its global index; the owner splices them into its
     partition's entity order with one gather on the device and reads the result back once.

With a warm start the prior models of the travelling entities go with them (coefficient counts, global feature
indices, coefficients); they start on the host (the model table) and are mapped to the solve's local order on the
host of the solving rank, so this small side channel is staged through page-locked memory.

Nothing here touches the arithmetic, but an entity is solved in ANOTHER BATCH than its partition alone would have formed, and
two of the device's routing thresholds are chosen per batch by default (where the one-CU tall variant and the four-workgroup
teams start; those variants add an entity's sums in different orders). A model with rebalance_entities therefore pins the
routing to the entity's own size (REDeviceSolver.pin_routing: fixed split, a team for every entity of at least 8 192 samples):
results are then bit-identical to an unbalanced run with the same pinned routing, whatever the plan moved
(tests/test_rebalance.py), and equal to a default-routed run's to rounding (<= 1e-7, on the tall entities only).
In the CPU tests the same code runs on CPU tensors over gloo.
"""
import numpy as np


# ---- planning (pure numpy, identical on every rank) -----------------------------------------------------------------
def plan_transfers(loads, tolerance=0.05):
    """loads[r] = total cost on rank r -> T[i, j] = cost rank i should send to rank j. Ranks within
    (1 + tolerance) x mean neither send nor receive."""
    loads = np.asarray(loads, np.float64)
    R = loads.size
    T = np.zeros((R, R))
    mean = loads.mean() if R else 0.0
    surplus = loads - mean
    give = np.where(surplus > tolerance * mean, surplus, 0.0)
    take = np.where(surplus < 0, -surplus, 0.0)
    donors = [int(i) for i in np.argsort(-give, kind="stable") if give[i] > 0]
    for i in donors:
        while give[i] > 1e-12:
            j = int(np.argmax(take))
            if take[j] <= 1e-12:
                break
            amt = min(give[i], take[j])
            T[i, j] += amt
            give[i] -= amt
            take[j] -= amt
    return T


def choose_entities(cost, amounts, order=None):
    """Which local entities go to each destination: entities in ascending cost order (ties by index) — or in the order
    `order` lists them, which may leave entities out: those never travel — are dealt out until each destination's amount
    is reached. Returns a list of index arrays, one per destination (possibly empty); every entity appears at most once;
    the rest stays."""
    cost = np.asarray(cost, np.float64)
    order = np.argsort(cost, kind="stable") if order is None else np.asarray(order, np.int64)
    csum = np.concatenate([[0.0], np.cumsum(cost[order])])   # costs are non-zero counts: the sums are exact
    out = []
    pos = 0
    for amt in amounts:
        end = pos
        if amt > 0 and pos < order.size:
            # the longest run order[pos:end] whose total stays within the amount
            end = int(np.searchsorted(csum, csum[pos] + amt * 1.0000001, side="right")) - 1
            end = min(max(end, pos), order.size)
        out.append(np.sort(order[pos:end].astype(np.int64)))
        pos = end
    return out


class CostModel:
    """Measured cost of an entity, in milliseconds of solve-kernel time: what the re-balancer equalises when the plain
    non-zero count is a poor proxy (a Zipf partition: the 1 % of entities that run on the team kernels cost several times
    more per non-zero than the small ones). Built from a solve that was timed per size class (gdmix_re_set_timing /
    gdmix_re_last_solve_ms): a class launch of `ms` milliseconds over entities holding `nnz` non-zeros prices a non-zero of
    that class at ms / nnz; rates are summed over ranks (`totals` -> all-reduce -> `from_totals`) so that every rank prices
    an entity the same. `additive` classes are those whose launch time is about the sum of its entities' work (one entity
    per wavefront group / workgroup); in the multi-workgroup team tiers the longest entity is the critical path, so moving
    their entities is not priced — they stay (`movable`)."""

    def __init__(self, rate, additive):
        self.rate = np.asarray(rate, np.float64)          # [classes] ms per non-zero (0 = class not seen)
        self.additive = np.asarray(additive, bool)

    @staticmethod
    def totals(cls, nnz, class_ms, num_classes):
        """This rank's [2, classes] sums: milliseconds and non-zeros per class (to be added up over ranks)."""
        cls = np.asarray(cls, np.int64)
        z = np.bincount(cls, weights=np.asarray(nnz, np.float64) + 64.0, minlength=num_classes)[:num_classes]
        return np.stack([np.asarray(class_ms, np.float64)[:num_classes], z])

    @classmethod
    def from_totals(cls, totals, additive):
        ms, z = np.asarray(totals, np.float64)
        # a class that ran inside another class's launch has entities and no time of its own: priced like its neighbour
        rate = np.where((z > 0) & (ms > 0), ms / np.maximum(z, 1.0), 0.0)
        seen = np.flatnonzero(rate > 0)
        if seen.size:
            for c in np.flatnonzero((z > 0) & (rate == 0)):
                rate[c] = rate[seen[np.argmin(np.abs(seen - c))]]
        return cls(rate, additive)

    def cost(self, cls, nnz):
        return self.rate[np.asarray(cls, np.int64)] * (np.asarray(nnz, np.float64) + 64.0)

    def order(self, cls, nnz):
        """The order in which entities are offered for travel: only entities of additive classes; the ones that cost most per
        byte moved first (the fewest bytes for the milliseconds shed), ties by size then index."""
        cls = np.asarray(cls, np.int64)
        ok = np.flatnonzero(self.additive[cls])
        key = np.lexsort((ok, np.asarray(nnz)[ok], -self.rate[cls[ok]]))
        return ok[key]


class SizeCostModel:
    """CostModel for entities that have not been classified yet (the next partition of a running job): the measured class times
    of the partitions solved so far, attributed to their entities by non-zeros and summed per SIZE bucket (floor(log2(nnz + 1))),
    price a non-zero of an entity of that size. Entities at or above `team_nnz` (the multi-workgroup team tiers: their launch time
    is the longest entity's critical path, not a sum) are priced but never offered for travel. `totals` of all ranks are added up
    (all-reduce) so that every rank prices an entity the same; a bucket nobody has seen yet takes its nearest neighbour's rate.
    Before any measurement exists the cost is the non-zero count (rate 1 everywhere)."""
    BUCKETS = 40

    def __init__(self, rate=None, team_nnz=16384):
        self.rate = np.ones(self.BUCKETS) if rate is None else np.asarray(rate, np.float64)
        self.team_nnz = int(team_nnz)

    @classmethod
    def bucket(cls, nnz):
        return np.minimum(np.floor(np.log2(np.asarray(nnz, np.float64) + 1.0)).astype(np.int64), cls.BUCKETS - 1)

    @classmethod
    def totals(cls, ent_cls, nnz, class_ms, num_classes):
        """[2, BUCKETS]: milliseconds and non-zeros per size bucket of one timed solve (entity classes from the solve, per-class ms)."""
        ent_cls = np.asarray(ent_cls, np.int64)
        w = np.asarray(nnz, np.float64) + 64.0
        per_class = np.bincount(ent_cls, weights=w, minlength=num_classes)[:num_classes]
        ms = np.asarray(class_ms, np.float64)[:num_classes]
        # a class without a launch of its own ran inside a neighbour's: its entities share that launch's time
        share = np.where(per_class > 0, ms / np.maximum(per_class, 1.0), 0.0)
        cost = share[ent_cls] * w
        b = cls.bucket(nnz)
        return np.stack([np.bincount(b, weights=cost, minlength=cls.BUCKETS), np.bincount(b, weights=w, minlength=cls.BUCKETS)])

    @classmethod
    def from_totals(cls, totals, team_nnz=16384):
        ms, z = np.asarray(totals, np.float64)
        rate = np.where((z > 0) & (ms > 0), ms / np.maximum(z, 1.0), 0.0)
        seen = np.flatnonzero(rate > 0)
        if seen.size == 0:
            return cls(None, team_nnz)
        for k in np.flatnonzero(rate == 0):
            rate[k] = rate[seen[np.argmin(np.abs(seen - k))]]
        return cls(rate, team_nnz)

    def cost(self, nnz):
        return self.rate[self.bucket(nnz)] * (np.asarray(nnz, np.float64) + 64.0)

    def order(self, nnz):
        nnz = np.asarray(nnz)
        ok = np.flatnonzero(nnz < self.team_nnz)
        return ok[np.lexsort((ok, nnz[ok], -self.rate[self.bucket(nnz[ok])]))]


def _segments(t, starts, lens, total):
    """Indices of the concatenated ranges [starts[k], starts[k] + lens[k]) as an int64 tensor on the tensors' device; `total`
    = sum(lens), known on the host (no device synchronisation)."""
    if total == 0:
        return t.zeros(0, dtype=t.int64, device=starts.device)
    out_start = t.cumsum(lens, 0) - lens
    return t.repeat_interleave(starts - out_start, lens, output_size=total) + t.arange(total, dtype=t.int64, device=starts.device)


class _Comm:
    """Collectives of the exchange over torch.distributed (nccl = RCCL: device tensors; gloo: CPU tensors)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.t, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        nccl = dist.get_backend(group) == "nccl"
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
        self.device = device
        # device tensors over gloo (ranks sharing one GPU in the harness tests — RCCL refuses two ranks on one device): the
        # collectives run on host copies. Never the case in a real job (nccl).
        self.staged = (not nccl) and device.type != "cpu"
        self.coll_device = torch.device("cpu") if self.staged else device
        self._pinned = {}     # (direction, dtype) -> page-locked staging tensor, grown as needed and reused from round to round
        self.bytes_sent = self.bytes_received = 0      # payload of all_to_all_t since construction (diagnostics)

    def _stage(self, key, count, tdt):
        """A page-locked host tensor of at least `count` elements (RCCL path: the device copies run at PCIe rate and without a
        pageable bounce buffer)."""
        buf = self._pinned.get((key, tdt))
        if buf is None or buf.numel() < count:
            buf = self.t.empty(max(count, 1) + max(count, 1) // 4, dtype=tdt).pin_memory()
            self._pinned[(key, tdt)] = buf
        return buf[:count]

    def all_gather_row(self, values):
        """values: a few floats of this rank -> [world, len(values)] numpy (small metadata)."""
        t = self.t
        mine = t.tensor([float(v) for v in values], dtype=t.float64, device=self.coll_device)
        out = [t.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.group)
        return np.stack([o.cpu().numpy() for o in out])

    def all_gather_floats(self, x):
        return self.all_gather_row([x])[:, 0]

    def exchange_counts(self, counts):
        """counts: [world, k] integers this rank sends to every rank -> [world, k] received from every rank."""
        t = self.t
        send = t.tensor(np.asarray(counts, np.int64).reshape(self.world, -1), dtype=t.int64, device=self.coll_device)
        recv = t.zeros_like(send)
        self.dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)
        return recv.cpu().numpy()

    def all_to_all_t(self, send, send_counts, recv_counts):
        """send: a tensor on the communicator's device, its rows grouped by destination rank (send_counts rows each; trailing
        dimensions travel along) -> the rows received, grouped by source rank (recv_counts rows each). Device to device."""
        t = self.t
        send = send.contiguous()
        if self.staged:
            send = send.cpu()
        recv = t.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=self.coll_device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=[int(c) for c in recv_counts],
                                    input_split_sizes=[int(c) for c in send_counts], group=self.group)
        row = send.element_size() * int(np.prod(send.shape[1:], dtype=np.int64))
        mine = self.rank
        self.bytes_sent += row * (int(sum(send_counts)) - int(send_counts[mine]))
        self.bytes_received += row * (int(sum(recv_counts)) - int(recv_counts[mine]))
        return recv.to(self.device) if self.staged else recv

    def all_to_all(self, parts, dtype):
        """parts[j] = numpy array for rank j -> list of numpy arrays received from every rank. Host data (the prior models of a
        warm start): staged through page-locked memory on the RCCL path."""
        t = self.t
        tdt = {np.int64: t.int64, np.float32: t.float32, np.float64: t.float64, np.uint8: t.uint8, np.int32: t.int32}[dtype]
        counts = [int(p.size) for p in parts]
        rs = [int(x) for x in self.exchange_counts(np.asarray(counts).reshape(-1, 1))[:, 0]]
        flat = np.concatenate([np.asarray(p, dtype) for p in parts]) if parts else np.zeros(0, dtype)
        on_device = self.device.type != "cpu"
        if on_device:
            hs = self._stage("send", flat.size, tdt)
            hs.numpy()[...] = flat
            send = hs.to(self.device, non_blocking=True)
        else:
            send = t.from_numpy(np.ascontiguousarray(flat))
        recv = self.all_to_all_t(send, counts, rs)
        if on_device:
            hr = self._stage("recv", recv.numel(), tdt)
            hr.copy_(recv, non_blocking=True)
            t.cuda.current_stream(self.device).synchronize()
            r = hr.numpy().copy()
        else:
            r = recv.numpy()
        out, p = [], 0
        for n in rs:
            out.append(r[p:p + n])
            p += n
        return out


WIRE_KEYS = ("ent_n", "row_nnz", "col_global", "val", "y", "offset", "weight")


def wire_tensors(batch, device, solver=None):
    """The partition in the exchange's wire form (gdmix_re_wire_batch with 32-bit counts, int32 feature ids, float labels) as
    tensors on `device`. With a device solver the arrays go up through its page-locked staging (upload_wire)."""
    import torch
    from .batch import WireRawBatch
    if isinstance(batch, WireRawBatch):      # narrowed (and range-checked) by the reader: only the widths are made uniform over the ranks
        n = batch.to_wire()
        w = dict(n, row_nnz=n["row_nnz"].astype(np.int32), row_nnz_width=4, col_global=n["col_global"].astype(np.int32), col_width=4,
                 y=batch.y, y_width=4)
    else:
        if batch.Z and (batch.col_global.min() < 0 or batch.col_global.max() > 0x7fffffff):
            raise ValueError("feature index outside [0, 2^31)")
        w = dict(E=batch.E, N=batch.N, Z=batch.Z, ent_n=np.diff(batch.ent_row_ptr).astype(np.int32), row_nnz=np.diff(batch.row_nnz_ptr).astype(np.int32),
                 row_nnz_width=4, col_global=batch.col_global.astype(np.int32), col_width=4, val=batch.val, y=batch.y, y_width=4,
                 offset=batch.offset, weight=batch.weight)
    if solver is not None and hasattr(solver, "upload_wire"):
        return solver.upload_wire(w, pinned=solver.wire_stage(w) if hasattr(solver, "wire_stage") else None)
    d = {k: w[k] for k in ("E", "N", "Z", "row_nnz_width", "y_width", "col_width")}
    for k in WIRE_KEYS:
        d[k] = None if w[k] is None else torch.from_numpy(np.ascontiguousarray(w[k])).to(device)
    return d


def wire_to_raw(work):
    """The wire form exchange() returns (CPU tensors) as a host RawBatch without ids (the CPU tests' solver stand-in takes host
    batches; the device solver widens the wire form in HBM instead: gdmix_re_widen)."""
    from .batch import RawBatch
    g = lambda k: None if work[k] is None else work[k].cpu().numpy()
    n, k = g("ent_n").astype(np.int64), g("row_nnz").astype(np.int64)
    return RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(n)]), row_nnz_ptr=np.concatenate([[0], np.cumsum(k)]),
                    col_global=g("col_global").astype(np.int64), val=g("val"), y=g("y"), offset=g("offset"), weight=g("weight"),
                    uid=None, entity_ids=None, has_label=True, trusted=True)


class Rebalancer:
    """One instance per training round. Usage on every rank (collective):

        rb = Rebalancer(ent_n, ent_nnz, wire)   # this rank's partition: samples / non-zeros per entity (host), wire form (tensors)
        work = rb.exchange()                    # wire form of what is solved here: kept + received entities (same device)
        ... widen, pack, solve `work` -> per-entity results in work's entity order, as tensors on the same device ...
        mine = rb.give_back(coef_cnt, theta, variance|None, feat_idx, ints, floats)   # results of MY entities, in my entity order

    A rank without a partition in a round passes empty arrays and still joins the collectives."""

    def __init__(self, ent_n, ent_nnz, wire, group=None, tolerance=0.05, cost=None, order=None, comm=None):
        """cost: per-entity cost the ranks equalise (default: non-zeros; CostModel.cost for measured milliseconds); order: the
        order in which this rank's entities are offered for travel (default: ascending cost; entities left out never travel;
        CostModel.order); comm: a _Comm kept from round to round (its page-locked staging blocks are reused)."""
        self.comm = comm if comm is not None else _Comm(group, wire["val"].device)
        self.t = self.comm.t
        self.n = np.asarray(ent_n, np.int64)
        self.nnz = np.asarray(ent_nnz, np.int64)
        self.E = int(self.n.size)
        self.wire = wire
        self.cost = np.asarray(self.nnz if cost is None else cost, np.float64)
        self.order = None if order is None else np.asarray(order, np.int64)
        self.tolerance = tolerance
        self.sent = None        # per destination: entity indices of this rank's partition
        self.kept = None
        self.recv_counts = None

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _gather_entities(self, ents):
        """Wire arrays of the entities `ents` (numpy indices into this rank's partition, any order), gathered on the device."""
        t, w = self.t, self.wire
        dev = w["val"].device
        e = t.from_numpy(np.ascontiguousarray(ents, np.int64)).to(dev)
        ent_n = w["ent_n"].to(t.int64)
        if not hasattr(self, "_row_start"):
            self._row_start = t.cumsum(ent_n, 0) - ent_n
            rz = w["row_nnz"].to(t.int64)
            self._nz_start = t.cumsum(rz, 0) - rz
        n_tot, z_tot = int(self.n[ents].sum()), int(self.nnz[ents].sum())
        rows = _segments(t, self._row_start[e], ent_n[e], n_tot)
        rz = w["row_nnz"][rows].to(t.int64)
        nz = _segments(t, self._nz_start[rows], rz, z_tot)
        return dict(ent_n=w["ent_n"][e], row_nnz=w["row_nnz"][rows], col_global=w["col_global"][nz], val=w["val"][nz], y=w["y"][rows],
                    offset=w["offset"][rows], weight=None if w["weight"] is None else w["weight"][rows])

    # -- the exchange ----------------------------------------------------------------------------------------------
    def exchange(self, prior=None, with_prior=False):
        """-> the wire form (dict for REDeviceSolver.widen; tensors on the communicator's device) of the entities solved here:
        this rank's kept entities in their order, then the received ones by origin rank. with_prior (the same on every rank):
        prior models travel too — prior = dict(has [E] bool, coef_ptr [E+1], theta, feat_ptr [E+1], idx) for this rank's
        partition in entity order (None = no entity has one); afterwards self.work_prior is the same structure for the
        returned batch."""
        c, t = self.comm, self.t
        has_w = self.wire["weight"] is not None
        meta = c.all_gather_row([self.cost.sum(), 1.0 if has_w else 0.0])
        self.loads = meta[:, 0]
        any_w = bool(meta[:, 1].any())
        if any_w and not has_w:     # a rank without weights may receive weighted entities (and vice versa): make the pieces uniform
            self.wire = dict(self.wire, weight=t.ones(self.wire["y"].numel(), dtype=t.float32, device=self.wire["val"].device))
        T = plan_transfers(self.loads, self.tolerance)
        self.sent = choose_entities(self.cost, T[c.rank], self.order)
        self.sent[c.rank] = np.zeros(0, np.int64)
        moving = np.concatenate(self.sent) if self.sent else np.zeros(0, np.int64)
        mask = np.ones(self.E, bool)
        mask[moving] = False
        self.kept = np.flatnonzero(mask)
        # rows of every array that go to each destination: entities, samples, non-zeros
        send_enz = np.array([[ix.size, int(self.n[ix].sum()), int(self.nnz[ix].sum())] for ix in self.sent], np.int64)
        recv_enz = c.exchange_counts(send_enz)
        self.recv_counts = [int(x) for x in recv_enz[:, 0]]
        out_part = self._gather_entities(moving) if moving.size else None
        kept_part = self.wire if moving.size == 0 else self._gather_entities(self.kept)
        level = {"ent_n": 0, "row_nnz": 1, "col_global": 2, "val": 2, "y": 1, "offset": 1, "weight": 1}
        work = {}
        for k in WIRE_KEYS:
            if k == "weight" and not any_w:
                work[k] = None
                continue
            lv = level[k]
            like = kept_part[k]
            send = out_part[k] if out_part is not None else like[:0]
            recv = c.all_to_all_t(send, send_enz[:, lv], recv_enz[:, lv])
            work[k] = t.cat([like, recv]) if recv.numel() else like
        work["E"] = int(self.kept.size + recv_enz[:, 0].sum())
        work["N"] = int(self.n[self.kept].sum() + recv_enz[:, 1].sum())
        work["Z"] = int(self.nnz[self.kept].sum() + recv_enz[:, 2].sum())
        work.update(row_nnz_width=4, y_width=4, col_width=4)
        self.work_E = work["E"]
        # the partition's own wire arrays and the prefix sums of the gathers are not needed again: when entities moved, `work`
        # holds copies of what stayed (the re-balanced path kept the partition in HBM several times over)
        self.wire = None
        for k in ("_row_start", "_nz_start"):
            if hasattr(self, k):
                delattr(self, k)
        self.work_prior = self._exchange_prior(prior) if with_prior else None
        return work

    def _exchange_prior(self, prior):
        """Prior models of the kept entities followed by those received, origin ranks in rank order (= the entity order
        of the batch exchange() returns)."""
        from .batch import _ranges
        c = self.comm
        E = self.E
        if prior is None:
            prior = dict(has=np.zeros(E, bool), coef_ptr=np.zeros(E + 1, np.int64), theta=np.zeros(0), feat_ptr=np.zeros(E + 1, np.int64),
                         idx=np.zeros(0, np.int64))
        has = np.asarray(prior["has"], bool)
        cp, fp = np.asarray(prior["coef_ptr"], np.int64), np.asarray(prior["feat_ptr"], np.int64)
        th, ix = np.asarray(prior["theta"], np.float64), np.asarray(prior["idx"], np.int64)

        def rows(sel):
            cc, fc = np.diff(cp)[sel], np.diff(fp)[sel]
            return has[sel].astype(np.int64), cc, fc, th[_ranges(cp[sel], cc)], ix[_ranges(fp[sel], fc)]
        i_parts, f_parts = [], []
        for sel in self.sent:
            h, cc, fc, tt, i = rows(sel)
            i_parts.append(np.concatenate([h, cc, fc, i]) if sel.size else np.zeros(0, np.int64))
            f_parts.append(tt if sel.size else np.zeros(0, np.float64))
        ri = c.all_to_all(i_parts, np.int64)
        rf = c.all_to_all(f_parts, np.float64)
        hs, ccs, fcs, ths, ixs = [[x] for x in rows(self.kept)]
        for j in range(c.world):
            n = self.recv_counts[j]
            if n == 0:
                continue
            ii = ri[j]
            hs.append(ii[:n]); ccs.append(ii[n:2 * n]); fcs.append(ii[2 * n:3 * n]); ixs.append(ii[3 * n:]); ths.append(rf[j])
        cc, fc = np.concatenate(ccs), np.concatenate(fcs)
        return dict(has=np.concatenate(hs).astype(bool), coef_ptr=np.concatenate([[0], np.cumsum(cc)]).astype(np.int64),
                    theta=np.concatenate(ths).astype(np.float64), feat_ptr=np.concatenate([[0], np.cumsum(fc)]).astype(np.int64),
                    idx=np.concatenate(ixs).astype(np.int64))

    def give_back(self, coef_cnt, theta, variance, feat_idx, ints, floats, has_intercept=True):
        """Per-entity results of the batch exchange() returned, in its entity order, as tensors on the communicator's device:
        coef_cnt [E'] (integers), theta [sum coef_cnt] float64, variance (same shape) or None, feat_idx [sum (coef_cnt - ic)]
        (integers: global feature index per non-intercept coefficient), ints [E', a] / floats [E', b]: per-entity solver
        statistics. Returns (coef_cnt, theta, variance, feat_idx, ints, floats) for THIS rank's partition in its entity
        order, on the same device. Device to device; the only host traffic is 2 x world counts."""
        c, t = self.comm, self.t
        dev = theta.device
        ic = 1 if has_intercept else 0
        cc = coef_cnt.to(t.int64)
        fc = cc - ic
        nk = int(self.kept.size)
        # foreign entities sit after the kept ones, grouped by origin rank in rank order
        seg = np.concatenate([[nk], nk + np.cumsum(self.recv_counts)]).astype(np.int64)
        cptr = t.cat([t.zeros(1, dtype=t.int64, device=dev), t.cumsum(cc, 0)])
        fptr = t.cat([t.zeros(1, dtype=t.int64, device=dev), t.cumsum(fc, 0)])
        at = t.from_numpy(seg).to(dev)
        cb, fb = cptr[at].cpu().numpy(), fptr[at].cpu().numpy()       # coefficient / feature offsets of the segment boundaries
        send_e = np.diff(seg)
        send_c, send_f = np.diff(cb), np.diff(fb)
        recv = c.exchange_counts(np.stack([send_e, send_c, send_f], axis=1))
        sent_e = np.array([ix.size for ix in self.sent], np.int64)
        if not np.array_equal(recv[:, 0], sent_e):
            raise RuntimeError("re-balancing: result counts differ from the entities sent")
        r_cc = c.all_to_all_t(cc[nk:], send_e, recv[:, 0])
        r_int = c.all_to_all_t(ints[nk:], send_e, recv[:, 0])
        r_flt = c.all_to_all_t(floats[nk:], send_e, recv[:, 0])
        r_th = c.all_to_all_t(theta[int(cb[0]):], send_c, recv[:, 1])
        r_va = c.all_to_all_t(variance[int(cb[0]):], send_c, recv[:, 1]) if variance is not None else None
        r_fi = c.all_to_all_t(feat_idx[int(fb[0]):], send_f, recv[:, 2])
        if nk == self.E and int(recv[:, 0].sum()) == 0 and int(send_e.sum()) == 0:
            # nothing left this rank and nothing came: its results already are in its partition's order (no copies: a rank
            # that stays out of a round pays nothing for it but the count exchanges)
            return cc, theta, variance, feat_idx, ints, floats
        # the pool: kept entities, then what every rank returned (in the order the entities were sent); an entity of the
        # partition finds its pool position through `where`
        where = np.empty(self.E, np.int64)
        where[self.kept] = np.arange(nk)
        pos = nk
        for ix in self.sent:
            where[ix] = pos + np.arange(ix.size)
            pos += ix.size
        w_t = t.from_numpy(where).to(dev)
        p_cc = t.cat([cc[:nk], r_cc])
        p_cstart = t.cumsum(p_cc, 0) - p_cc
        p_fc = p_cc - ic
        p_fstart = t.cumsum(p_fc, 0) - p_fc
        my_cc = p_cc[w_t]
        total_c = int(cb[0]) + int(recv[:, 1].sum())
        total_f = int(fb[0]) + int(recv[:, 2].sum())
        cg = _segments(t, p_cstart[w_t], my_cc, total_c)
        fg = _segments(t, p_fstart[w_t], p_fc[w_t], total_f)
        th_all = t.cat([theta[:int(cb[0])], r_th])
        va_out = t.cat([variance[:int(cb[0])], r_va])[cg] if variance is not None else None
        fi_all = t.cat([feat_idx[:int(fb[0])], r_fi])
        return (my_cc, th_all[cg], va_out, fi_all[fg], t.cat([ints[:nk], r_int])[w_t], t.cat([floats[:nk], r_flt])[w_t])
