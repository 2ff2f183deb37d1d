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
