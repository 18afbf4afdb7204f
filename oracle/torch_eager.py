"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Op-for-op torch-CPU eager restatement of one BPRLoss.stageOne of the reference, used by
bench.py's cpu_baseline leg ("a torch-CPU eager variant that mirrors
model.py:201-231 / utils.py:53-64") and cross-checked against oracle/lgcn_oracle.c in
tests/test_oracle.py.  It issues the same library calls in the same order as the reference
does on its CPU path:

    model.py:209      torch.cat([users_emb, items_emb])
    model.py:216-218  K x torch.sparse.mm(A_hat_coo, x)
    model.py:221-222  torch.stack(...).mean(dim=1), split
    model.py:131-133  three row gathers
    model.py:168-173  mul/sum, logsigmoid, mean, three norm(2).pow(2)
    utils.py:56-64    loss + decay*reg, zero_grad, backward, Adam.step, loss.cpu().item()

Parity status: pinned through oracle/lgcn_oracle.c (same inputs, losses agree to 3e-6), which
is itself pinned by the fixtures captured from the reference (oracle/oracle.py header).
"""
import numpy as np
import torch
import torch.nn.functional as F


class EagerTrainer:
    def __init__(self, n_users, indptr, indices, vals, e0, K, decay=1e-4, lr=1e-3, threads=None, reg_rows='propagated'):
        assert reg_rows in ('propagated', 'ego')      # 'ego': upstream LightGCN's L2 term (userEmb0 / posEmb0 / negEmb0), see lgcn_oracle.c orc_bpr_ego
        self.reg_rows = reg_rows
        if threads:
            torch.set_num_threads(int(threads))
        N = len(indptr) - 1
        rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(np.asarray(indptr, np.int64)))
        ij = torch.from_numpy(np.vstack([rows, np.asarray(indices, np.int64)]))
        # dataloader.py:183-190,244: coalesced fp32 COO with int64 indices
        self.graph = torch.sparse_coo_tensor(ij, torch.from_numpy(np.asarray(vals, np.float32)), (N, N)).coalesce()
        e0 = torch.from_numpy(np.ascontiguousarray(e0, np.float32))
        self.n_users, self.K, self.decay = int(n_users), int(K), float(decay)
        self.user_w = torch.nn.Parameter(e0[:n_users].clone())
        self.item_w = torch.nn.Parameter(e0[n_users:].clone())
        self.opt = torch.optim.Adam([self.user_w, self.item_w], lr=lr)          # utils.py:51

    def computer(self):
        x = torch.cat([self.user_w, self.item_w])
        layers = [x]
        for _ in range(self.K):
            x = torch.sparse.mm(self.graph, x)
            layers.append(x)
        out = torch.stack(layers, dim=1).mean(dim=1)
        return torch.split(out, [self.n_users, out.shape[0] - self.n_users])

    def stageOne(self, users, pos, neg):
        users, pos, neg = (torch.as_tensor(np.asarray(t), dtype=torch.long) for t in (users, pos, neg))
        all_users, all_items = self.computer()
        u, p, n = all_users[users], all_items[pos], all_items[neg]
        bpr = -torch.mean(F.logsigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)))
        if self.reg_rows == 'ego':
            u0, p0, n0 = self.user_w[users], self.item_w[pos], self.item_w[neg]
            reg = 0.5 * (u0.norm(2).pow(2) + p0.norm(2).pow(2) + n0.norm(2).pow(2)) / float(len(users))
        else:
            reg = 0.5 * (u.norm(2).pow(2) + p.norm(2).pow(2) + n.norm(2).pow(2)) / float(len(users))
        loss = bpr + self.decay * reg
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.cpu().item()

    @property
    def e0(self):
        return torch.cat([self.user_w, self.item_w]).detach().numpy()
