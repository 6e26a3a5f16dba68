"""CPU stand-in for dist.HipShardBackend built on the oracle (TESTS ONLY): lets the sharded frame logic and
its collectives run under gloo without a GPU.  The product never imports this."""
import math

import numpy as np
import torch

from oracle import oracle as orc

BLOCK = 4096


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

    def tail_a(self, st, softmax):
        nb = st.nb
        x = st.scores.numpy()[st.nn_idx.numpy()]
        e = np.array([math.exp(v) for v in (x - 1.0)])  # glibc exp, constant shift 1
        valid = st.valid.numpy().astype(bool)
        st.e.copy_(torch.as_tensor(e))
        st.x.copy_(torch.as_tensor(x))
        em, xm = e * valid, x * valid
        for b in range(nb):
            sl = slice(b * BLOCK, (b + 1) * BLOCK)
            st.r1[b] = orc.blocked_scan(e[sl])[1]
            lp, tot = orc.blocked_scan(em[sl])
            st.cdf[sl] = torch.as_tensor(lp)
            st.r1[nb + b] = tot
            lpr, totr = orc.blocked_scan(xm[sl])
            st.lp_raw[sl] = torch.as_tensor(lpr)
            st.r1[2 * nb + b] = totr
            st.r1[3 * nb + b], st.r1[4 * nb + b] = float(x[sl].max()), float(x[sl].min())
        apply_local = bool(softmax)
        st.r1[5 * nb] = float(np.isnan(em if apply_local else xm).sum())
        st.r1[5 * nb + 1] = float(valid.sum())

    def tail_fin(self, st, r1_all, rank, world, n_total, softmax, want_rmse):
        nb = st.nb
        r1 = r1_all.numpy().reshape(world, 5 * nb + 4)
        mx, mn = float(r1[:, 3 * nb:4 * nb].max()), float(r1[:, 4 * nb:5 * nb].min())
        apply = bool(softmax) and not (abs(mx - mn) <= 1e-8)
        S = _seq_sum(r1[:, :nb].reshape(-1)) if apply else 1.0
        valid = st.valid.numpy().astype(bool)
        ev = st.e.numpy() if apply else st.x.numpy()
        st.weights.copy_(torch.as_tensor((ev / S) * valid))
        tot = (r1[:, nb:2 * nb] if apply else r1[:, 2 * nb:3 * nb]).reshape(-1)
        total = _seq_sum(tot)
        lp = (st.cdf if apply else st.lp_raw).numpy().copy()
        out = np.empty_like(lp)
        for b in range(nb):
            bp = _seq_sum(tot[: rank * nb + b])
            out[b * BLOCK:(b + 1) * BLOCK] = (bp + lp[b * BLOCK:(b + 1) * BLOCK]) / total if total != 0 else np.nan
        if rank == world - 1:
            out[-1] = 1.0
        st.cdf.copy_(torch.as_tensor(out))
        status = 2 if r1[:, 5 * nb].sum() != 0 else 0
        if np.isnan(total):
            status |= 2
        elif total == 0.0:
            status |= 1
        st.status[0] = status
        st.status[1] = int(r1[:, 5 * nb + 1].sum())
        if want_rmse:
            st.rmse[0] = float(np.sqrt(r1[:, 5 * nb + 2].sum() / n_total))
            st.rmse[1] = float(np.sqrt(r1[:, 5 * nb + 3].sum() / n_total))

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
