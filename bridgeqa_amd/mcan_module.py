"""MCAN blocks used by the reference-object head of ScanQA's VQA branch -- mirror of the reference's
models/mcan_module.py (FC :18-43, MLP :46-54, LayerNorm :57-69, MHAtt :138-224, FFN :229-245, SA :250-273, SGA :278-311)
with its class names, constructor arguments, forward signatures and state-dict keys (linear_v / linear_k / linear_q /
linear_merge, mlp.fc.linear, mlp.linear, a_2 / b_2), so reference checkpoints load with strict=True.

SURVEY.md §8f rank 1 ("ScanQA.forward glue"): these blocks sit on the CALLER side of the hot path -- 2 layers over 256
proposals x 256 channels against 20 question states, < 0.1 % of the step's flop -- so they run on torch ops (fp32), no
kernels of their own.  AttFlat and the MCAN_E / MCAN_ED stacks belong to the non-BLIP branch (qa_module.py:496-590) and are
not part of this path.

Reference quirks kept on purpose (parity, not opinion):
  * LayerNorm divides by (std + eps) with the UNBIASED standard deviation (torch.std), not sqrt(var + eps);
  * masks are boolean with True = "hide this key" (scores.masked_fill(mask, -1e9)).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class FC(nn.Module):
    def __init__(self, in_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.pdrop, self.use_gelu = pdrop, use_gelu
        self.linear = nn.Linear(in_size, out_size)
        if use_gelu:
            self.gelu = nn.GELU()
        if pdrop > 0:
            self.dropout = nn.Dropout(pdrop)

    def forward(self, x):
        x = self.linear(x)
        if self.use_gelu:
            x = self.gelu(x)
        return self.dropout(x) if self.pdrop > 0 else x


class MLP(nn.Module):
    def __init__(self, in_size, mid_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.fc = FC(in_size, mid_size, pdrop=pdrop, use_gelu=use_gelu)
        self.linear = nn.Linear(mid_size, out_size)

    def forward(self, x):
        return self.linear(self.fc(x))


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std + eps) + b_2 with torch.std's unbiased estimate (mcan_module.py:57-69)"""

    def __init__(self, size, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.a_2 = nn.Parameter(torch.ones(size))
        self.b_2 = nn.Parameter(torch.zeros(size))

    def forward(self, x):
        centred = x - x.mean(-1, keepdim=True)
        n = x.shape[-1]
        std = centred.pow(2).sum(-1, keepdim=True).div(max(n - 1, 1)).sqrt()
        return self.a_2 * centred / (std + self.eps) + self.b_2


class MHAtt(nn.Module):
    def __init__(self, hidden_size, num_heads=8, pdrop=0.1):
        super().__init__()
        self.linear_v = nn.Linear(hidden_size, hidden_size)
        self.linear_k = nn.Linear(hidden_size, hidden_size)
        self.linear_q = nn.Linear(hidden_size, hidden_size)
        self.linear_merge = nn.Linear(hidden_size, hidden_size)
        self.hidden_size, self.num_heads = hidden_size, num_heads
        self.head_hidden_size = int(hidden_size / num_heads)
        self.dropout = nn.Dropout(pdrop)

    def _split(self, t, B):
        return t.view(B, -1, self.num_heads, self.head_hidden_size).transpose(1, 2)

    def forward(self, v, k, q, mask, att_pdrop=0, att_drop_topk=100):
        B = q.size(0)
        v, k, q = self._split(self.linear_v(v), B), self._split(self.linear_k(k), B), self._split(self.linear_q(q), B)
        ctx = self.att(v, k, q, mask, att_pdrop, att_drop_topk)
        return self.linear_merge(ctx.transpose(1, 2).contiguous().view(B, -1, self.hidden_size))

    def att(self, value, key, query, mask, att_pdrop=0, att_drop_topk=100):
        # (att_pdrop / att_drop_topk: accepted and ignored, as in the reference -- its top-k masking is commented out)
        scores = torch.matmul(query, key.transpose(-2, -1)) / math.sqrt(query.size(-1))
        if mask is not None:
            scores = scores.masked_fill(mask, -1e9)
        return torch.matmul(self.dropout(F.softmax(scores, dim=-1)), value)


class FFN(nn.Module):
    def __init__(self, hidden_size, pdrop=0.1):
        super().__init__()
        self.mlp = MLP(in_size=hidden_size, mid_size=int(hidden_size * 4), out_size=hidden_size, pdrop=pdrop, use_gelu=True)

    def forward(self, x):
        return self.mlp(x)


class SA(nn.Module):
    """self-attention block, post-norm"""

    def __init__(self, hidden_size, num_heads=8, pdrop=0.1):
        super().__init__()
        self.mhatt = MHAtt(hidden_size, num_heads, pdrop)
        self.ffn = FFN(hidden_size, pdrop)
        self.dropout1, self.norm1 = nn.Dropout(pdrop), LayerNorm(hidden_size)
        self.dropout2, self.norm2 = nn.Dropout(pdrop), LayerNorm(hidden_size)

    def forward(self, x, x_mask):
        x = self.norm1(x + self.dropout1(self.mhatt(x, x, x, x_mask)))
        return self.norm2(x + self.dropout2(self.ffn(x)))


class SGA(nn.Module):
    """self-attention over x, then x attends to y (guided attention), then FFN; post-norm"""

    def __init__(self, hidden_size, num_heads=8, pdrop=0.1):
        super().__init__()
        self.mhatt1 = MHAtt(hidden_size, num_heads, pdrop)
        self.mhatt2 = MHAtt(hidden_size, num_heads, pdrop)
        self.ffn = FFN(hidden_size, pdrop)
        self.dropout1, self.norm1 = nn.Dropout(pdrop), LayerNorm(hidden_size)
        self.dropout2, self.norm2 = nn.Dropout(pdrop), LayerNorm(hidden_size)
        self.dropout3, self.norm3 = nn.Dropout(pdrop), LayerNorm(hidden_size)

    def forward(self, x, y, x_mask, y_mask, att_pdrop=None, att_drop_topk=None):
        x = self.norm1(x + self.dropout1(self.mhatt1(x, x, x, x_mask)))
        x = self.norm2(x + self.dropout2(self.mhatt2(y, y, x, y_mask, att_pdrop, att_drop_topk)))
        return self.norm3(x + self.dropout3(self.ffn(x)))
