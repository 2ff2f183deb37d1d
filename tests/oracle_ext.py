"""Test-only backend: the nine oracle ops + CPU restatements of the point-major
extras (plain torch indexing on top of the same pinned arithmetic), shaped like
``pointnet2_ops._ext`` so the product's python layer can run on CPU tensors in
the ``-m "not gpu"`` suite and serve as the checker in the GPU parity tests."""
import torch

from oracle.oracle import OracleExt


def _bidx(B, *rest):
    return torch.arange(B).view(B, *([1] * len(rest)))


class OracleRowsExt(OracleExt):
    HAS_ROWS = True

    @staticmethod
    def group_concat_rows(xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius):
        B, m, ns = idx.shape
        li = idx.long()
        b = torch.arange(B).view(B, 1, 1)
        parts = []
        if use_xyz:
            rel = xyz[b, li] - new_xyz.unsqueeze(2)          # (B,m,ns,3)
            if normalize:
                rel = rel / torch.tensor(radius, dtype=torch.float32)
            parts.append(rel)
        if feats_rows is not None:
            parts.append(feats_rows[b, li])                   # (B,m,ns,C)
        return torch.cat(parts, dim=3).contiguous()

    @staticmethod
    def group_rows_grad(grad_out, idx, n, c, col0):
        B, m, ns, W = grad_out.shape
        out = torch.zeros(B, n, c, dtype=torch.float32)
        g = grad_out[..., col0:col0 + c].reshape(B, m * ns, c)
        out.scatter_add_(1, idx.long().view(B, m * ns, 1).expand(-1, -1, c), g)
        return out

    @staticmethod
    def rows_max(x):
        out, arg = x.max(dim=1)
        # torch.max returns *an* arg-max; make it the first one like max_pool2d
        first = (x == out.unsqueeze(1)).float().argmax(dim=1)
        return out.contiguous(), first.to(torch.int32).contiguous()

    @staticmethod
    def rows_max_grad(grad_out, arg, ns):
        R, C = grad_out.shape
        gx = torch.zeros(R, ns, C, dtype=torch.float32)
        gx.scatter_(1, arg.long().unsqueeze(1), grad_out.unsqueeze(1))
        return gx

    @staticmethod
    def three_interpolate_rows(feats_rows, idx, weight, out=None, col0=0):
        # via the channel-major oracle op so the pinned fma shape is reused
        ch = OracleExt.three_interpolate(feats_rows.transpose(1, 2).contiguous(), idx.contiguous(),
                                         weight.contiguous())
        rows = ch.transpose(1, 2).contiguous()
        if out is None:
            return rows
        out[..., col0:col0 + rows.size(2)] = rows
        return out

    @staticmethod
    def three_interpolate_rows_grad(grad_out, idx, weight, m, c, col0=0):
        g = grad_out[..., col0:col0 + c].transpose(1, 2).contiguous()
        return OracleExt.three_interpolate_grad(g, idx, weight, m).transpose(1, 2).contiguous()

    @staticmethod
    def gather_rows(x, index, out=None, col0=0, check=True):
        rows = OracleExt.gather_rows(x.contiguous(), index.contiguous())
        if out is None:
            return rows
        out[:, col0:col0 + rows.size(1)] = rows
        return out

    @staticmethod
    def scatter_add_rows(src, index, dim_size, h=None, col0=0, check=True):
        h = src.size(1) if h is None else h
        return OracleExt.scatter_add_rows(src[:, col0:col0 + h].contiguous(), index.contiguous(), dim_size)

    @staticmethod
    def segment_sum_rows(src, order, rowptr, dim_size, h=None, col0=0):
        h = src.size(1) if h is None else h
        s = src[:, col0:col0 + h]
        out = torch.zeros(dim_size, h, dtype=torch.float32)
        for n in range(dim_size):
            for p in range(int(rowptr[n]), int(rowptr[n + 1])):
                out[n] += s[int(order[p])]
        return out

    @staticmethod
    def ball_query_unique_resample(idx, seed, want_cnt=True):
        """GF3D/pointnet2/pointnet2_utils.py:327-336 restated on CPU: torch.unique per region for the leading part and
        the count (the reference's own ops), the tail drawn with the product's counter-based generator
        (csrc/ball_query.hip pn2_mix32) instead of torch.randint so that the result is comparable bit for bit."""
        B, m, ns = idx.shape
        cnt = torch.zeros(B, m)
        M = 0xFFFFFFFF
        for b in range(B):
            for r in range(m):
                uniq = torch.unique(idx[b, r, :])
                n = uniq.shape[0]
                cnt[b, r] = n
                row = b * m + r
                tail = []
                for s_ in range(n, ns):
                    h = (seed ^ ((row * 0x9E3779B9) & M) ^ ((s_ * 0x85EBCA6B) & M)) & M
                    h ^= h >> 16; h = (h * 0x7FEB352D) & M
                    h ^= h >> 15; h = (h * 0x846CA68B) & M
                    h ^= h >> 16
                    tail.append(int(uniq[h % n]))
                idx[b, r, :] = torch.cat([uniq, torch.tensor(tail, dtype=idx.dtype)])
        return cnt if want_cnt else None

    @staticmethod
    def segment_bn_rows(x, ptr, gamma, beta, eps, relu, h=None, col0=0):
        """What the reference's one-scan-per-step loop computes: torch's own batch_norm (batch statistics) on the rows of
        each scan separately (network_TripletGCN.py:20 — track_running_stats=False), then the optional ReLU."""
        import torch.nn.functional as F
        C = x.size(1) if h is None else h
        xs = x[:, col0:col0 + C]
        S = ptr.numel() - 1
        y = torch.empty(x.size(0), C)
        mean, rstd = torch.empty(S, C), torch.empty(S, C)
        for s_ in range(S):
            a, b = int(ptr[s_]), int(ptr[s_ + 1])
            seg = xs[a:b]
            out = F.batch_norm(seg, None, None, gamma, beta, True, 0.0, eps)
            y[a:b] = F.relu(out) if relu else out
            mean[s_] = seg.mean(0)
            rstd[s_] = 1.0 / torch.sqrt(seg.var(0, unbiased=False) + eps)
        return y, mean, rstd

    @staticmethod
    def segment_bn_rows_grad(grad_out, x, ptr, gamma, beta, mean, rstd, relu, col0=0, eps=1e-5):
        """Backward of the above through torch autograd, scan by scan."""
        import torch.nn.functional as F
        C = grad_out.size(1)
        S = ptr.numel() - 1
        gx = torch.empty(x.size(0), C)
        dg, db = torch.zeros(C), torch.zeros(C)
        for s_ in range(S):
            a, b = int(ptr[s_]), int(ptr[s_ + 1])
            with torch.enable_grad():                   # called from inside an autograd Function's backward
                seg = x[a:b, col0:col0 + C].detach().clone().requires_grad_(True)
                ga, be = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
                out = F.batch_norm(seg, None, None, ga, be, True, 0.0, eps)
                out = F.relu(out) if relu else out
                out.backward(grad_out[a:b])
            gx[a:b] = seg.grad
            dg += ga.grad
            db += be.grad
        return gx, dg, db

    @staticmethod
    def gather2_add_rows(q, p, ia, ib, cola, colb):
        H = q.size(1)
        q += p[ia, cola:cola + H] + p[ib, colb:colb + H]
        return q

    @staticmethod
    def segment_sum2_rows(src, order, rowptr, dim_size, h, col0, col1):
        s = (src[:, col0:col0 + h] + src[:, col1:col1 + h]).contiguous()
        return OracleRowsExt.segment_sum_rows(s, order, rowptr, dim_size)
