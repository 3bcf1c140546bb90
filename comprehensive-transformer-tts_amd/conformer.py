"""`conformer` block plugin (reference: model/transformers/conformer.py) on the gfx950 kernels.

Plugin contract (CompTransTTS.py:19-39): `TextEncoder(config).forward(tokens, pad_mask) -> (enc, word_emb)`,
`Decoder(config).forward(x, pad_mask) -> (dec, mask)`, both expose `.d_model`; state-dict keys follow
SURVEY.md Appendix A (`layer_stack.{i}.sequential.{0..4}...`, the sinusoid table re-registered per layer).

Behaviour kept bug-compatible with the reference: the attention never sees the padding
mask (`nn.Sequential` drops it, conformer.py:243 vs :326) and scales scores by sqrt(d_model) (:375,:409); only the
block output is zeroed at pads; the decoder crops to max_seq_len in training (:148-154).

Per block the GEMM-shaped work (FF linears, q/k/v/pos/out projections, both pointwise convs, QK^T / QP^T / PV and
their gradients) runs on ctts_gemm; GLU, depthwise conv k=31, BatchNorm+Swish, the relative shift + softmax + dropout
are streaming kernels (csrc/conformer.hip, csrc/norm.hip).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .configs import N_SYMBOLS
from .model import _Linear, _Norm, _BatchNorm, _Conv


def interleaved_sinusoid_table(n_position, d_hid):
    """blocks.py:26-46: angle = pos / 10000^(2*(j//2)/d); sin on even channels, cos on odd (float64 -> float32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab = np.array(ang)
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).float()


class _LinearNorm(nn.Module):                      # key: <name>.linear.{weight,bias}
    def __init__(self, cin, cout, bias=False):
        super().__init__()
        self.linear = _Linear(cin, cout, bias=bias)


class _PointConv(nn.Module):                       # key: <name>.conv.{weight[Cout,Cin,1],bias}
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(cin, cout, 1)


class _DepthConv(nn.Module):                       # key: <name>.conv.weight[C,1,k]   (no bias)
    def __init__(self, c, k):
        super().__init__()
        self.conv = nn.Module()
        self.conv.weight = nn.Parameter(torch.empty(c, 1, k))


def _seq(**children):
    m = nn.Module()
    for k, v in children.items():
        m.add_module(k.lstrip("_"), v)
    return m


class _Residual(nn.Module):                        # key: sequential.{i}.module....
    def __init__(self, module):
        super().__init__()
        self.module = module


class _RelAttention(nn.Module):
    def __init__(self, d_model, n_heads):
        super().__init__()
        self.u_bias = nn.Parameter(torch.empty(n_heads, d_model // n_heads))
        self.v_bias = nn.Parameter(torch.empty(n_heads, d_model // n_heads))
        self.query_proj = _LinearNorm(d_model, d_model)
        self.key_proj = _LinearNorm(d_model, d_model)
        self.value_proj = _LinearNorm(d_model, d_model)
        self.pos_proj = _LinearNorm(d_model, d_model)
        self.out_proj = _LinearNorm(d_model, d_model)


class _MHSAModule(nn.Module):
    def __init__(self, d_model, n_heads, position_enc):
        super().__init__()
        self.positional_encoding = position_enc      # the SAME Parameter object as <stack>.position_enc (conformer.py:322)
        self.layer_norm = _Norm(d_model)
        self.attention = _RelAttention(d_model, n_heads)


class ConformerBlock(nn.Module):
    """conformer.py:162-246"""

    def __init__(self, d_model, n_heads, ff_factor, conv_factor, ksize, dropout, half_step, position_enc):
        super().__init__()
        assert conv_factor == 2, "Currently, Only Supports expansion_factor 2"          # conformer.py:451
        assert (ksize - 1) % 2 == 0
        self.d_model, self.n_heads, self.dropout = d_model, n_heads, dropout
        self.ff_factor = 0.5 if half_step else 1.0

        def ff():
            return _Residual(_seq(sequential=_seq(_0=_Norm(d_model), _1=_LinearNorm(d_model, d_model * ff_factor, True),
                                                  _4=_LinearNorm(d_model * ff_factor, d_model, True))))
        conv = _Residual(_seq(sequential=_seq(_0=_Norm(d_model), _2=_PointConv(d_model, 2 * d_model), _4=_DepthConv(d_model, ksize),
                                              _5=_BatchNorm(d_model), _7=_PointConv(d_model, d_model))))
        self.sequential = _seq(_0=ff(), _1=_Residual(_MHSAModule(d_model, n_heads, position_enc)), _2=conv, _3=ff(),
                               _4=_Norm(d_model))
        self.drop_ctx = None

    def _ff(self, x, res, p, drop):
        s = res.module.sequential
        ln, l1, l2 = getattr(s, "0"), getattr(s, "1").linear, getattr(s, "4").linear
        h, xr = ops.layer_norm_res(x, ln.weight, ln.bias, 1e-5)
        link = ops.EpiLink()       # Swish' and the dropout mask of the first linear ride in the second one's data-gradient GEMM
        h = ops.linear(h, l1.weight, l1.bias, act=ops.ACT_SWISH, p_drop=p, drop=drop, link=link, link_role=1)
        return ops.linear(h, l2.weight, l2.bias, alpha=self.ff_factor, residual=xr, p_drop=p, drop=drop, link=link, link_role=2)

    def forward(self, x, nonpad, pos_table):
        """x [B,T,C]; nonpad float [B*T]; pos_table [T,C] (rows of the sinusoid table)"""
        B, T, C = x.shape
        p = self.dropout if self.training else 0.0
        drop = self.drop_ctx if p > 0 else None
        seq = self.sequential
        x = self._ff(x, getattr(seq, "0"), p, drop)
        # ---- relative-position multi-head self-attention (mask deliberately NOT applied, conformer.py:243)
        m = getattr(seq, "1").module
        at = m.attention
        h, xr = ops.layer_norm_res(x, m.layer_norm.weight, m.layer_norm.bias, 1e-5)
        # one [768,256] GEMM for q | k | v; the three weights are neighbours in model.parameters() order, so in the flat parameter /
        # gradient arenas the stacked matrix and its gradient are views (no cat, no split-and-add in the backward)
        qkv = ops.linear_packed(h, (at.query_proj.linear.weight, at.key_proj.linear.weight, at.value_proj.linear.weight))
        qu, qv, kv = ops.relattn_split(qkv, at.u_bias, at.v_bias)        # q + u_bias, q + v_bias, k | v in one pass
        pos = ops.linear(pos_table, at.pos_proj.linear.weight)           # [T,C], batch independent
        ctxv = ops.relpos_attention(qu, qv, kv, pos, self.n_heads, 1.0 / math.sqrt(C), p_drop=p, drop=drop)
        x = ops.linear(ctxv, at.out_proj.linear.weight, None, residual=xr, p_drop=p, drop=drop)
        # ---- convolution module
        s = getattr(seq, "2").module.sequential
        ln, pw1, dw, bn, pw2 = getattr(s, "0"), getattr(s, "2").conv, getattr(s, "4").conv, getattr(s, "5"), getattr(s, "7").conv
        h, xr = ops.layer_norm_res(x, ln.weight, ln.bias, 1e-5)
        h = ops.linear(h, pw1.weight.view(2 * C, C), pw1.bias)
        h = ops.glu(h)
        h = ops.depthwise_conv1d(h, dw.weight)
        h = ops.batch_norm_act(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, self.training,
                               act=ops.ACT_SWISH)
        x = ops.linear(h, pw2.weight.view(C, C), pw2.bias, residual=xr, p_drop=p, drop=drop)
        x = self._ff(x, getattr(seq, "3"), p, drop)
        fin = getattr(seq, "4")
        return ops.layer_norm(x, fin.weight, fin.bias, 1e-5, rowscale=nonpad)   # LN then masked_fill(pad, 0)


class _ConformerStack(nn.Module):
    def __init__(self, config, which):
        super().__init__()
        c = config["conformer"]
        self.d_model = c[f"{which}_hidden"]
        self.max_seq_len = config["max_seq_len"]
        self.position_enc = nn.Parameter(interleaved_sinusoid_table(self.max_seq_len + 1, self.d_model).unsqueeze(0),
                                         requires_grad=False)
        if which == "encoder":      # registered BEFORE layer_stack like the reference (conformer.py:41-48): `parameters()` order is what
            self.src_word_emb = nn.Embedding(N_SYMBOLS + 1, self.d_model, padding_idx=0)     # index-based Adam state is keyed on
        self.layer_stack = nn.ModuleList([
            ConformerBlock(self.d_model, c[f"{which}_head"], c["feed_forward_expansion_factor"], c["conv_expansion_factor"],
                           c["conv_kernel_size"], c[f"{which}_dropout"], c["half_step_residual"], self.position_enc)
            for _ in range(c[f"{which}_layer"])])

    def _pos(self, T, device):
        if T > self.max_seq_len:
            if self.training:
                raise ValueError(f"sequence length {T} exceeds max_seq_len {self.max_seq_len}")
            return interleaved_sinusoid_table(T, self.d_model).to(device)      # conformer.py:73-78,140-145
        return self.position_enc[0, :T]

    def run(self, x, mask, pos_table):
        from .model import mask_aux
        nonpad = mask_aux(mask)[0]
        pos_table = pos_table.contiguous()
        cut = getattr(self, "_cut_prefix", None)
        for li, blk in enumerate(self.layer_stack):
            if cut is not None:
                x = ops.stage_cut(x, f"{cut}.{li}")
            x = blk(x, nonpad, pos_table)
        return x


class TextEncoder(_ConformerStack):
    """conformer.py:20-88"""

    def __init__(self, config):
        super().__init__(config, "encoder")

    def forward(self, src_seq, mask):
        emb = ops.embedding(src_seq, self.src_word_emb.weight, 0)
        pos = self._pos(src_seq.shape[1], emb.device)
        return self.run(emb + pos.unsqueeze(0), mask, pos), emb


class Decoder(_ConformerStack):
    """conformer.py:91-159"""

    def __init__(self, config):
        super().__init__(config, "decoder")

    def forward(self, enc_seq, mask):
        T = enc_seq.shape[1]
        if not (not self.training and T > self.max_seq_len):
            T = min(T, self.max_seq_len)                     # crop (conformer.py:148-154)
            enc_seq, mask = enc_seq[:, :T, :], mask[:, :T]
        pos = self._pos(T, enc_seq.device)
        return self.run(enc_seq + pos.unsqueeze(0), mask, pos), mask


def reset_conformer_parameters(stack):
    """initialisers of the reference: xavier_uniform for LinearNorm weights and u/v biases (blocks.py:159-172,
    conformer.py:367-368), torch defaults for the Conv1d layers, nn.Embedding default N(0,1) with zero pad row."""
    params = dict(stack.named_parameters())
    for name, p in params.items():
        if name.endswith("position_enc") or name.endswith("positional_encoding"):
            continue
        if name.endswith(".linear.weight") or name.endswith("u_bias") or name.endswith("v_bias"):
            nn.init.xavier_uniform_(p)
        elif name.endswith(".linear.bias"):
            nn.init.zeros_(p)
        elif name.endswith("src_word_emb.weight"):
            nn.init.normal_(p)
            with torch.no_grad():
                p[0].zero_()
        elif name.endswith("conv.weight"):
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
        elif name.endswith("conv.bias"):
            fan_in = params[name[:-4] + "weight"][0].numel()
            nn.init.uniform_(p, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
