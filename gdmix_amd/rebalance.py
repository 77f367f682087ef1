"""Entity re-balancing across the GPUs of a node for skewed partitions.

The reference shards by partition only (worker w trains partitions[w::W], random_effect_driver.py:60-68): with a
Zipf-distributed entity size one partition — hence one worker — can hold most of the work. Entities are
independent, so any entity may be solved on any rank; only its result has to come back to the rank that owns its
partition, which writes the model file. One round per partition step, all ranks in lockstep:

  1. all_gather of every rank's total cost (cost of an entity = its non-zeros: what the solver streams);
  2. every rank derives the same transfer matrix from the gathered loads (deterministic greedy: largest surplus
     to largest deficit) and picks which of its own entities travel — smallest first, so the giants, which no
     rank could absorb, stay where they are;
  3. one all_to_all of the raw ragged arrays of the travelling entities (three flat buffers: int64, float32, id
     bytes); on the GPU box the tensors live in HBM and the collective is RCCL over xGMI (direct, non-ring
     all-to-all: 7 links per GPU), in the tests it is gloo on CPU;
     with a warm start the prior models of the travelling entities go with them (coefficient counts, global
     feature indices, coefficients: one more int64 and one float64 all_to_all);
  4. every rank solves its kept + received entities as one batch;
  5. one all_to_all back: coefficient counts, coefficients, variances, feature indices, solver statistics of the
     foreign entities; the owner splices them into its results in the original entity order.

Nothing here touches the arithmetic: results are bit-identical to the unbalanced run (tests/test_distributed.py).
"""
import numpy as np

from .batch import RawBatch, concat
from .io.native_reader import _ids_to_bytes, _split_ids   # vectorised id <-> bytes (pure numpy; no library call)


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


def choose_entities(cost, amounts):
    """Which local entities go to each destination: entities in ascending cost order (ties by index) are dealt out
    until each destination's amount is reached. Returns a list of index arrays, one per destination (possibly
    empty); every entity appears at most once; the rest stays."""
    cost = np.asarray(cost, np.float64)
    order = np.argsort(cost, kind="stable")
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


# ---- wire format of a RawBatch slice ----------------------------------------------------------------------------------
def _pack(b: RawBatch):
    """RawBatch -> (int64 array, float32 array, uint8 array)."""
    id_bytes, id_ptr = _ids_to_bytes(b.entity_ids)
    id_len = np.diff(id_ptr)
    has_w = b.weight is not None
    i64 = np.concatenate([np.array([b.E, b.N, b.Z, int(has_w), int(b.has_label)], np.int64), np.diff(b.ent_row_ptr),
                          np.diff(b.row_nnz_ptr), b.col_global, b.uid if b.uid is not None else np.zeros(b.N, np.int64), id_len])
    f32 = np.concatenate([b.val, b.y, b.offset] + ([b.weight] if has_w else []))
    u8 = np.frombuffer(id_bytes, np.uint8)
    return i64.astype(np.int64), f32.astype(np.float32), u8


def _unpack(i64, f32, u8):
    E, N, Z, has_w, has_label = (int(x) for x in i64[:5])
    p = 5
    n = i64[p:p + E]; p += E
    k = i64[p:p + N]; p += N
    col = i64[p:p + Z]; p += Z
    uid = i64[p:p + N]; p += N
    id_len = i64[p:p + E]; p += E
    q = 0
    val = f32[q:q + Z]; q += Z
    y = f32[q:q + N]; q += N
    off = f32[q:q + N]; q += N
    w = f32[q:q + N] if has_w else None
    idp = np.concatenate([[0], np.cumsum(id_len)]).astype(np.int64)
    ids = _split_ids(u8.tobytes(), idp, E)
    return RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(n)]).astype(np.int64),
                    row_nnz_ptr=np.concatenate([[0], np.cumsum(k)]).astype(np.int64), col_global=col.copy(), val=val.copy(),
                    y=y.copy(), offset=off.copy(), weight=None if w is None else w.copy(), uid=uid.copy(), entity_ids=ids,
                    has_label=bool(has_label))


def _empty_like(b: RawBatch):
    return b.select(np.zeros(0, np.int64))


class _Comm:
    """all_to_all of variable-length numpy arrays over torch.distributed (nccl = RCCL: device tensors; gloo: CPU)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.t, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.device = device
        self._pinned = {}     # (direction, dtype) -> page-locked staging tensor, grown as needed and reused from round to round

    def _stage(self, key, count, tdt):
        """A page-locked host tensor of at least `count` elements (RCCL path: the device copies run at PCIe rate and without a
        pageable bounce buffer)."""
        buf = self._pinned.get((key, tdt))
        if buf is None or buf.numel() < count:
            buf = self.t.empty(max(count, 1) + max(count, 1) // 4, dtype=tdt).pin_memory()
            self._pinned[(key, tdt)] = buf
        return buf[:count]

    def all_gather_floats(self, x):
        t = self.t
        mine = t.tensor([float(x)], dtype=t.float64, device=self.device)
        out = [t.zeros(1, dtype=t.float64, device=self.device) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.group)
        return np.array([float(o.item()) for o in out])

    def all_to_all(self, parts, dtype):
        """parts[j] = numpy array for rank j -> list of arrays received from every rank."""
        t = self.t
        tdt = {np.int64: t.int64, np.float32: t.float32, np.float64: t.float64, np.uint8: t.uint8, np.int32: t.int32}[dtype]
        sizes = t.tensor([int(p.size) for p in parts], dtype=t.int64, device=self.device)
        rsizes = t.zeros(self.world, dtype=t.int64, device=self.device)
        self.dist.all_to_all_single(rsizes, sizes, group=self.group)
        rs = [int(x) for x in rsizes.tolist()]
        flat = np.concatenate([np.asarray(p, dtype) for p in parts]) if parts else np.zeros(0, dtype)
        on_device = self.device.type != "cpu"
        if on_device:   # host arrays -> page-locked staging -> HBM; the exchange itself is device to device (RCCL over xGMI)
            hs = self._stage("send", flat.size, tdt)
            hs.numpy()[...] = flat
            send = hs.to(self.device, non_blocking=True)
        else:
            send = t.from_numpy(np.ascontiguousarray(flat))
        recv = t.zeros(sum(rs), dtype=tdt, device=self.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=rs, input_split_sizes=[int(p.size) for p in parts],
                                    group=self.group)
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


class Rebalancer:
    """One instance per training round. Usage on every rank (collective):

        rb = Rebalancer(batch)                 # batch may be empty (rank without a partition this round)
        work = rb.exchange()                   # RawBatch to solve here: kept + received entities
        ... solve `work` -> per-entity results in work's entity order ...
        mine = rb.give_back(coef_cnt, theta, variance|None, feat_cnt, feat_idx, stats)   # results of MY entities,
                                                                                        # in batch's entity order
    """

    def __init__(self, batch: RawBatch, group=None, device=None, tolerance=0.05, cost=None):
        self.batch = batch
        self.comm = _Comm(group, device)
        self.cost = np.asarray(batch.ent_nnz() if cost is None else cost, np.float64)
        self.tolerance = tolerance
        self.sent = None        # per destination: indices into batch
        self.kept = None
        self.recv_counts = None

    def exchange(self, prior=None, with_prior=False) -> RawBatch:
        """with_prior (the same on every rank): prior models travel too. prior = dict(has [E] bool, coef_ptr [E+1],
        theta, feat_ptr [E+1], idx) for this rank's batch in entity order (None = no entity has one); afterwards
        self.work_prior is the same structure for the returned batch."""
        c = self.comm
        loads = c.all_gather_floats(self.cost.sum())
        self.loads = loads
        T = plan_transfers(loads, self.tolerance)
        self.sent = choose_entities(self.cost, T[c.rank])
        self.sent[c.rank] = np.zeros(0, np.int64)
        moving = np.concatenate(self.sent) if self.sent else np.zeros(0, np.int64)
        mask = np.ones(self.batch.E, bool)
        mask[moving] = False
        self.kept = np.flatnonzero(mask)
        packs = [_pack(self.batch.select(ix)) if ix.size else (np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.uint8))
                 for ix in self.sent]
        ri = c.all_to_all([p[0] for p in packs], np.int64)
        rf = c.all_to_all([p[1] for p in packs], np.float32)
        ru = c.all_to_all([p[2] for p in packs], np.uint8)
        parts = [self.batch.select(self.kept)]
        self.recv_counts = []
        for j in range(c.world):
            if ri[j].size:
                b = _unpack(ri[j], rf[j], ru[j])
                # a rank without weights may receive weighted entities (and vice versa): make the pieces uniform
                parts.append(b)
                self.recv_counts.append(b.E)
            else:
                self.recv_counts.append(0)
        has_w = any(p.weight is not None and p.E for p in parts)
        if has_w:
            for p in parts:
                if p.weight is None:
                    p.weight = np.ones(p.N, np.float32)
        self.work_prior = None
        if with_prior:
            self.work_prior = self._exchange_prior(prior)
        parts = [p for p in parts if p.E] or [parts[0]]
        self.work = concat(parts) if len(parts) > 1 else parts[0]
        self.work_has_label = all(p.has_label for p in parts)
        return self.work

    def _exchange_prior(self, prior):
        """Prior models of the kept entities followed by those received, origin ranks in rank order (= the entity order
        of the batch exchange() returns)."""
        from .batch import _ranges
        c = self.comm
        E = self.batch.E
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
            h, cc, fc, t, i = rows(sel)
            i_parts.append(np.concatenate([h, cc, fc, i]) if sel.size else np.zeros(0, np.int64))
            f_parts.append(t if sel.size else np.zeros(0, np.float64))
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

    def give_back(self, coef_cnt, theta, variance, feat_cnt, feat_idx, stats=None):
        """Per-entity results of the batch returned by exchange(), in its entity order:
        coef_cnt[E'], theta[sum coef_cnt] (float64), variance or None, feat_cnt[E'], feat_idx[sum feat_cnt] (int64),
        stats: dict of per-entity int32/float64 arrays (nit, nfev, status, fval, gnorm) or None.
        Returns the same tuple for THIS rank's original batch, in its original entity order."""
        c = self.comm
        coef_cnt = np.asarray(coef_cnt, np.int64)
        feat_cnt = np.asarray(feat_cnt, np.int64)
        cptr = np.concatenate([[0], np.cumsum(coef_cnt)])
        fptr = np.concatenate([[0], np.cumsum(feat_cnt)])
        nk = self.kept.size
        stats = stats or {}
        skeys = sorted(stats)
        # foreign entities sit after the kept ones, grouped by origin rank in rank order
        seg = [nk]
        for j in range(c.world):
            seg.append(seg[-1] + self.recv_counts[j])

        def slices(j):
            a, b = seg[j], seg[j + 1]
            return a, b, cptr[a], cptr[b], fptr[a], fptr[b]
        i64_parts, f64_parts = [], []
        for j in range(c.world):
            a, b, c0, c1, f0, f1 = slices(j)
            if b == a:
                i64_parts.append(np.zeros(0, np.int64))
                f64_parts.append(np.zeros(0, np.float64))
                continue
            ints = [np.array([b - a, int(variance is not None)], np.int64), coef_cnt[a:b], feat_cnt[a:b], np.asarray(feat_idx, np.int64)[f0:f1]]
            flts = [np.asarray(theta, np.float64)[c0:c1]]
            if variance is not None:
                flts.append(np.asarray(variance, np.float64)[c0:c1])
            for k in skeys:
                flts.append(np.asarray(stats[k], np.float64)[a:b])
            i64_parts.append(np.concatenate(ints))
            f64_parts.append(np.concatenate(flts))
        ri = c.all_to_all(i64_parts, np.int64)
        rf = c.all_to_all(f64_parts, np.float64)
        # assemble my batch's results in original entity order: every source (kept here, returned by rank j) is appended
        # to a pool, every entity remembers where its slice of the pool starts
        from .batch import _ranges
        E = self.batch.E
        theta = np.asarray(theta, np.float64)
        feat_idx = np.asarray(feat_idx, np.int64)
        my_coef = np.zeros(E, np.int64)
        my_feat = np.zeros(E, np.int64)
        c_start = np.zeros(E, np.int64)
        f_start = np.zeros(E, np.int64)
        any_var = variance is not None
        th_pool, fi_pool, va_pool = [theta[:cptr[nk]]], [feat_idx[:fptr[nk]]], [None if variance is None else np.asarray(variance, np.float64)[:cptr[nk]]]
        st_out = {k: np.zeros(E, np.float64) for k in skeys}
        my_coef[self.kept] = coef_cnt[:nk]
        my_feat[self.kept] = feat_cnt[:nk]
        c_start[self.kept] = cptr[:nk]
        f_start[self.kept] = fptr[:nk]
        for k in skeys:
            st_out[k][self.kept] = np.asarray(stats[k], np.float64)[:nk]
        c_base, f_base = int(cptr[nk]), int(fptr[nk])
        for j in range(c.world):
            ix = self.sent[j]
            if ix.size == 0:
                continue
            ii, ff = ri[j], rf[j]
            n, hv = int(ii[0]), int(ii[1])
            assert n == ix.size, "result count differs from the entities sent"
            cc = ii[2:2 + n]; fc = ii[2 + n:2 + 2 * n]; fi = ii[2 + 2 * n:]
            tot = int(cc.sum())
            q = tot
            va = None
            if hv:
                va = ff[q:q + tot]; q += tot
                any_var = True
            for k in skeys:
                st_out[k][ix] = ff[q:q + n]; q += n
            my_coef[ix] = cc
            my_feat[ix] = fc
            c_start[ix] = c_base + np.cumsum(cc) - cc
            f_start[ix] = f_base + np.cumsum(fc) - fc
            th_pool.append(ff[:tot]); fi_pool.append(fi); va_pool.append(va)
            c_base += tot
            f_base += int(fc.sum())
        cg = _ranges(c_start, my_coef)
        th_all = np.concatenate(th_pool).astype(np.float64)
        fi_all = np.concatenate(fi_pool).astype(np.int64)
        va_out = None
        if any_var:
            va_all = np.concatenate([v if v is not None else np.zeros(t.size) for v, t in zip(va_pool, th_pool)]).astype(np.float64)
            va_out = va_all[cg]
        return my_coef, th_all[cg], va_out, my_feat, fi_all[_ranges(f_start, my_feat)], st_out
