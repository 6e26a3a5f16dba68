"""CPU stand-in for dist.HipShardBackend built on the oracle (TESTS ONLY): lets the sharded frame logic and
its collectives run under gloo without a GPU.  The product never imports this."""
import math

import numpy as np
import torch

from oracle import oracle as orc

BLOCK = 4096
ROUTE_REC = 88


def _seq_sum(a):
    acc = 0.0
    for v in np.asarray(a, dtype=np.float64):
        acc = acc + float(v)
    return acc


class OracleShardBackend:
    device = torch.device("cpu")

    def __init__(self, cb_poses, cb_embeddings, mesh_vertices):
        self.ofl = orc.OracleFilter(cb_poses, cb_embeddings, mesh_vertices)
        self.cb_poses = np.asarray(cb_poses, dtype=np.float32)
        self.K = self.cb_poses.shape[0]

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def project(self, poses):
        idx = self.ofl.SE3_NN_idx(poses.numpy())
        return torch.as_tensor(self.cb_poses[idx]), torch.as_tensor(idx)

    def front(self, st, odom, code, gt, tn, rot, std_t, std_r, seed, step, prune_thr, use_hint=True, scores_ready=False):
        N, base = st.N, st.slot_base
        if tn is None:
            tn_all, rot_all = orc.philox_noise(base + N, seed, step, np.float32(std_t), np.float32(std_r))
            tn, rot = tn_all[base:], rot_all[base:]
        else:
            tn, rot = tn.numpy(), rot.numpy()
        p1 = orc.propagate(st.poses.numpy(), odom.numpy(), tn, rot)
        idx, _ = orc.nn6(orc.R3_SE3(p1), self.ofl.cb_feat)
        if not scores_ready:
            st.scores.copy_(torch.as_tensor(orc.score_codebook(self.ofl.emb, code.numpy())))
        dist = orc.nn3_dist(p1, self.ofl.verts)
        st.poses_prop.copy_(torch.as_tensor(p1))
        st.nn_idx.copy_(torch.as_tensor(idx))
        st.valid.copy_(torch.as_tensor((~(dist > prune_thr)).astype(np.uint8)))
        st.r1[5 * st.nb:] = 0.0
        if gt is not None:
            rt, rr = orc.particle_rmse(p1, gt.numpy())
            st.r1[5 * st.nb + 2], st.r1[5 * st.nb + 3] = rt * rt * N, rr * rr * N

    @staticmethod
    def _views(st):
        """e | x | lp | lp_raw views of the shard's tables block (the chunk / group tables behind them are unused here)."""
        N = st.N
        Np = -(-N // 16) * 16
        t = st.tables
        return t[0:N], t[Np:Np + N], t[2 * Np:2 * Np + N], t[3 * Np:3 * Np + N]

    def tail_a(self, st, softmax):
        nb = st.nb
        te, tx, tlp, tlpr = self._views(st)
        x = st.scores.numpy()[st.nn_idx.numpy()]
        e = orc.exp_spec(x, 1.0)  # the spec exponential, constant shift 1
        valid = st.valid.numpy().astype(bool)
        te.copy_(torch.as_tensor(e))
        tx.copy_(torch.as_tensor(x))
        em, xm = e * valid, x * valid
        for b in range(nb):
            sl = slice(b * BLOCK, (b + 1) * BLOCK)
            st.r1[b] = orc.blocked_scan(e[sl])[1]
            lp, tot = orc.blocked_scan(em[sl])
            tlp[sl] = torch.as_tensor(lp)
            st.r1[nb + b] = tot
            lpr, totr = orc.blocked_scan(xm[sl])
            tlpr[sl] = torch.as_tensor(lpr)
            st.r1[2 * nb + b] = totr
            st.r1[3 * nb + b], st.r1[4 * nb + b] = float(x[sl].max()), float(x[sl].min())
        st.r1[5 * nb] = float(np.isnan(em if softmax else xm).sum())
        st.r1[5 * nb + 1] = float(valid.sum())

    def _globals(self, st, r1_all, world, softmax):
        """guard, S, per-block totals of the chosen variant, total - what every rank derives from the gathered records"""
        nb = st.nb
        r1 = r1_all.numpy().reshape(world, 5 * nb + 4)
        mx, mn = float(r1[:, 3 * nb:4 * nb].max()), float(r1[:, 4 * nb:5 * nb].min())
        apply = bool(softmax) and not (abs(mx - mn) <= 1e-8)
        S = _seq_sum(r1[:, :nb].reshape(-1)) if apply else 1.0
        tot = (r1[:, nb:2 * nb] if apply else r1[:, 2 * nb:3 * nb]).reshape(-1)
        total = _seq_sum(tot)
        status = 2 if r1[:, 5 * nb].sum() != 0 else 0
        if np.isnan(total):
            status |= 2
        elif total == 0.0:
            status |= 1
        return r1, apply, S, tot, total, status

    def _finish(self, st, r1, status, n_total, want_rmse):
        nb = st.nb
        st.status[0] = status
        st.status[1] = int(r1[:, 5 * nb + 1].sum())
        if want_rmse:
            st.rmse[0] = float(np.sqrt(r1[:, 5 * nb + 2].sum() / n_total))
            st.rmse[1] = float(np.sqrt(r1[:, 5 * nb + 3].sum() / n_total))

    def _local_cdf(self, st, apply, tot, total, rank, world):
        te, tx, tlp, tlpr = self._views(st)
        lp = (tlp if apply else tlpr).numpy()
        out = np.empty_like(lp)
        for b in range(st.nb):
            bp = _seq_sum(tot[: rank * st.nb + b])
            out[b * BLOCK:(b + 1) * BLOCK] = (bp + lp[b * BLOCK:(b + 1) * BLOCK]) / total if total != 0 else np.nan
        if rank == world - 1:
            out[-1] = 1.0
        return out

    def tail_fin(self, st, r1_all, rank, world, n_total, softmax, want_rmse):
        r1, apply, S, tot, total, status = self._globals(st, r1_all, world, softmax)
        te, tx, _, _ = self._views(st)
        valid = st.valid.numpy().astype(bool)
        st.weights.copy_(torch.as_tensor(((te if apply else tx).numpy() / S) * valid))
        st.cdf.copy_(torch.as_tensor(self._local_cdf(st, apply, tot, total, rank, world)))
        self._finish(st, r1, status, n_total, want_rmse)

    # ---- owner-side resample: same protocol as HipShardBackend.route / unpack ---------------------------------
    def route(self, st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse):
        N, nb = st.N, st.nb
        n_all = world * N
        r1, apply, S, tot, total, status = self._globals(st, r1_all, world, softmax)
        self._finish(st, r1, status, n_all, want_rmse)
        te, tx, _, _ = self._views(st)
        valid = st.valid.numpy().astype(bool)
        w_loc = ((te if apply else tx).numpy() / S) * valid
        st.weights.copy_(torch.as_tensor(w_loc))
        slots = np.arange(n_all)
        dest = slots // N
        if status != 0:
            owner, src_loc = dest.copy(), slots - dest * N
        else:
            if mode == 0:
                t = u_all.numpy() if u_all is not None else orc.philox_uniform64(n_all, seed, step)
                side = "left"
            else:
                r = np.float32(u32) if u32 >= 0 else np.float32(orc.philox_uniform32(seed, step))
                off = np.float32(r / np.float32(n_all))
                t = np.arange(n_all, dtype=np.float64) / float(n_all) + float(off)
                t = np.where(t >= 1.0, t - 1.0, t)
                side = "right"
            # exact cdf at the block ends in global block order -> owner block; then the owner's own cdf
            bps = np.array([_seq_sum(tot[:b]) for b in range(world * nb)])
            ends = (bps + tot) / total
            ends[-1] = 1.0
            blk = np.searchsorted(ends, t, side=side)
            blk = np.minimum(blk, world * nb - 1)
            owner = blk // nb
            cdf_loc = self._local_cdf(st, apply, tot, total, rank, world)
            src_loc = np.minimum(np.searchsorted(cdf_loc, t, side=side), N - 1)  # meaningful where owner == rank
        mine = owner == rank
        sends = [int((mine & (dest == d)).sum()) for d in range(world)]
        recvs = [int(((dest == rank) & (owner == o)).sum()) for o in range(world)]
        rec = np.zeros((int(mine.sum()), ROUTE_REC), dtype=np.uint8)
        order = np.argsort(dest[mine], kind="stable")
        sl, sr = slots[mine][order], src_loc[mine][order]
        rec[:, 0:4] = (sl - dest[mine][order] * N).astype(np.int32).view(np.uint8).reshape(-1, 4)
        rec[:, 4:8] = (rank * N + sr).astype(np.int32).view(np.uint8).reshape(-1, 4)
        rec[:, 8:12] = st.nn_idx.numpy()[sr].astype(np.int32).view(np.uint8).reshape(-1, 4)
        rec[:, 16:24] = w_loc[sr].astype(np.float64).view(np.uint8).reshape(-1, 8)
        rec[:, 24:88] = st.poses_prop.numpy().reshape(N, 16)[sr].astype(np.float32).view(np.uint8).reshape(-1, 64)
        return torch.as_tensor(rec.reshape(-1)), sends, recvs

    def route_fixed(self, st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse, cap, ovf_cap):
        """The fixed-capacity form: the rows of `route`, destination by destination, into padded segments; what does not
        fit goes to the overflow block with the destination rank in the fourth int."""
        rec, sends, _ = self.route(st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse)
        rec = rec.numpy().reshape(-1, ROUTE_REC)
        send = np.full((world * cap, ROUTE_REC), 0xFF, dtype=np.uint8)
        ovf = np.full((ovf_cap, ROUTE_REC), 0xFF, dtype=np.uint8)
        o, q = 0, 0
        for d, n in enumerate(sends):
            rows = rec[o:o + n].copy()
            rows[:, 12:16] = np.array([d], dtype=np.int32).view(np.uint8)
            if d == rank:  # own slot, own source: never travels
                st._self_rows = rows
                o += n
                continue
            fit = min(n, cap)
            send[d * cap:d * cap + fit] = rows[:fit]
            extra = rows[fit:]
            assert q + len(extra) <= ovf_cap, "overflow block too small"
            ovf[q:q + len(extra)] = extra
            q += len(extra)
            o += n
        st.overflow_rows = q
        return torch.as_tensor(send.reshape(-1)), torch.as_tensor(ovf.reshape(-1))

    def unpack_fixed(self, st, recv, ovf_all, rank):
        rows = [recv.numpy().reshape(-1, ROUTE_REC)]
        o = ovf_all.numpy().reshape(-1, ROUTE_REC)
        rows.append(o[o[:, 12:16].copy().view(np.int32).reshape(-1) == rank])
        rows.append(st._self_rows)
        rec = np.concatenate(rows)
        rec = rec[rec[:, 0:4].copy().view(np.int32).reshape(-1) >= 0]
        self.unpack(st, torch.as_tensor(rec.reshape(-1)))

    def unpack(self, st, recv):
        rec = recv.numpy().reshape(-1, ROUTE_REC)
        assert rec.shape[0] == st.N
        slot = rec[:, 0:4].copy().view(np.int32).reshape(-1)
        assert np.array_equal(np.sort(slot), np.arange(st.N)), "every slot receives exactly one row"
        st.ridx[torch.as_tensor(slot).long()] = torch.as_tensor(rec[:, 4:8].copy().view(np.int32).reshape(-1))
        st.hint[torch.as_tensor(slot).long()] = torch.as_tensor(rec[:, 8:12].copy().view(np.int32).reshape(-1))
        st.weights_res[torch.as_tensor(slot).long()] = torch.as_tensor(rec[:, 16:24].copy().view(np.float64).reshape(-1))
        st.poses[torch.as_tensor(slot).long()] = torch.as_tensor(rec[:, 24:88].copy().view(np.float32).reshape(-1, 4, 4))

    def tail_resample(self, st, pack_all, n_all, mode, u, u32, seed, step):
        N, base, G = st.N, st.slot_base, n_all // st.N
        blocks = pack_all.reshape(G, st.stride)
        cdf_all = torch.cat([blocks[r, 0:8 * N].view(torch.float64) for r in range(G)])
        weights_all = torch.cat([blocks[r, 8 * N:16 * N].view(torch.float64) for r in range(G)])
        poses_all = torch.cat([blocks[r, 16 * N:80 * N].view(torch.float32).view(N, 4, 4) for r in range(G)])
        nn_all = torch.cat([blocks[r, 80 * N:84 * N].view(torch.int32) for r in range(G)])
        if int(st.status[0]) != 0:
            src = np.arange(base, base + N, dtype=np.int32)
        elif mode == 0:
            uu = u.numpy() if u is not None else orc.philox_uniform64(base + N, seed, step)[base:]
            src = orc.search_lower(cdf_all.numpy(), uu)
        else:
            r = u32 if u32 >= 0 else orc.philox_uniform32(seed, step)
            src = orc.search_systematic(cdf_all.numpy(), n_all, r)[base:base + N]
        st.ridx.copy_(torch.as_tensor(src))
        s = torch.as_tensor(src).long()
        st.poses.copy_(poses_all[s])
        st.weights_res.copy_(weights_all[s])
        st.hint.copy_(nn_all[s])
