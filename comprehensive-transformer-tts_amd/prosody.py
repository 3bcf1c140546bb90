"""liu2021 implicit prosody modelling (reference model/modules.py:332-648, model/coordconv.py:33-70,140-159)
on the gfx950 kernels.  State-dict keys follow the reference (incl. the unused `convs.0.weight/bias` that CoordConv2d inherits from
nn.Conv2d next to its real `convs.0.conv.*`, and torch's `gru.weight_ih_l0[_reverse]` names).

Layout: the Conv2d stack runs channel-last on x[B,T,W,C]; each layer is patch-matrix (csrc/prosody.hip) -> ctts_gemm -> BatchNorm2d
statistics over all B*T*W rows (pads included, as the reference) + ReLU fused in one apply kernel.  The reference flattens
[N,C,T,W'] -> [N,T,C*W'] channel-major; here the data stays [N,T,W',C] and the GRU's W_ih columns are permuted instead.
The GRU input projection is one GEMM over all time steps; only the h_{t-1} -> h_t recurrence is sequential (ctts_gru_fwd/bwd).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .model import _Linear, _Norm, _BatchNorm, _Conv, _ConvNormK


class _Conv2dParams(nn.Module):
    """nn.Conv2d parameters with the reference's shape [Cout, Cin, 3, 3] (state-dict / optimizer compatible) but GEMM-major MEMORY
    [Cout][kh][kw][Cin] behind permuted strides (model._Conv does the same for Conv1d): the patch-matrix GEMM reads the weight as is
    and the weight-gradient GEMM accumulates straight into `.grad` - no permuted copy per layer and call, no accumulate-add launch."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, 3, 3, cin).permute(0, 3, 1, 2))
        self.bias = nn.Parameter(torch.empty(cout))


class _CoordConv2dParams(_Conv2dParams):
    """coordconv.py:140-159: the subclass keeps nn.Conv2d's own (unused) weight/bias and convolves with `self.conv`."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout)
        self.conv = _Conv2dParams(cin + 3, cout)        # + xx, yy, rr (rank 2, with_r=True)


class _GRUParams(nn.Module):
    def __init__(self, idim, hidden, bidirectional):
        super().__init__()
        self.hidden = hidden
        self.bidirectional = bidirectional
        for sfx in ([""] + (["_reverse"] if bidirectional else [])):
            setattr(self, "weight_ih_l0" + sfx, nn.Parameter(torch.empty(3 * hidden, idim)))
            setattr(self, "weight_hh_l0" + sfx, nn.Parameter(torch.empty(3 * hidden, hidden)))
            setattr(self, "bias_ih_l0" + sfx, nn.Parameter(torch.empty(3 * hidden)))
            setattr(self, "bias_hh_l0" + sfx, nn.Parameter(torch.empty(3 * hidden)))

    def run(self, x, w_ih=None):
        w_ih = self.weight_ih_l0 if w_ih is None else w_ih
        if not self.bidirectional:
            return ops.gru(x, w_ih, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0)
        return ops.gru(x, w_ih, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0, self.weight_ih_l0_reverse,
                       self.weight_hh_l0_reverse, self.bias_ih_l0_reverse, self.bias_hh_l0_reverse)


def coord_planes(T, W, device):
    """AddCoords(rank=2, with_r=True) planes (coordconv.py:33-70): xx varies along T (dim_y), yy along W (dim_x), both in [-1,1];
    rr = sqrt((xx-.5)^2 + (yy-.5)^2).  T is the padded batch length (bug-compatible: the planes depend on the padding)."""
    xx = torch.arange(T, dtype=torch.int32, device=device)[:, None].expand(T, W).float() / (T - 1)
    yy = torch.arange(W, dtype=torch.int32, device=device)[None, :].expand(T, W).float() / (W - 1)
    xx = xx * 2 - 1
    yy = yy * 2 - 1
    rr = torch.sqrt(torch.pow(xx - 0.5, 2) + torch.pow(yy - 0.5, 2))
    return torch.stack([xx, yy, rr], -1)                  # [T,W,3]


class ReferenceEncoder(nn.Module):
    """modules.py:332-397"""

    def __init__(self, preprocess_config, model_config):
        super().__init__()
        cfg = model_config["prosody_modeling"]["liu2021"]
        self.n_mel = preprocess_config["preprocessing"]["mel"]["n_mel_channels"]
        filters = [1] + list(cfg["ref_enc_filters"])
        if list(cfg["ref_enc_size"]) != [3, 3] or list(cfg["ref_enc_strides"]) != [1, 2] or list(cfg["ref_enc_pad"]) != [1, 1]:
            raise NotImplementedError("ReferenceEncoder kernels are built for ref_enc_size [3,3], strides [1,2], pad [1,1]")
        K_ = len(cfg["ref_enc_filters"])
        self.convs = nn.ModuleList([_CoordConv2dParams(filters[0], filters[1])]
                                   + [_Conv2dParams(filters[i], filters[i + 1]) for i in range(1, K_)])
        self.bns = nn.ModuleList([_BatchNorm(filters[i + 1]) for i in range(K_)])
        w = self.n_mel
        for _ in range(K_):
            w = (w - 3 + 2) // 2 + 1
        self.out_w, self.out_c = w, filters[-1]
        self.gru = _GRUParams(filters[-1] * w, cfg["ref_enc_gru_size"], False)
        self._coords = None

    def forward(self, mel, nonpad_rows):
        """mel [N,T,n_mel] (device), nonpad_rows float [N*T] -> memory [N,T,G] (G = gru size); last state = memory[:, -1]"""
        gi, w_hh, b_hh = self.features(mel, nonpad_rows)
        return ops.gru_group([gi], [w_hh], [b_hh])[0]

    def features(self, mel, nonpad_rows):
        """conv stack + GRU input projection: -> (gi [N,T,3G], W_hh, b_hh).  The recurrence itself is left to the caller so that the
        two reference encoders of the variance adaptor share ONE launch (ops.gru_group)."""
        N, T, W = mel.shape
        if self._coords is None or self._coords.shape[:2] != (T, W) or self._coords.device != mel.device:
            self._coords = coord_planes(T, W, mel.device)
        x = torch.cat([mel.unsqueeze(-1), self._coords.unsqueeze(0).expand(N, T, W, 3)], -1)      # [N,T,W,4] = mel|xx|yy|rr
        for i, (cv, bn) in enumerate(zip(self.convs, self.bns)):
            p = cv.conv if i == 0 else cv
            x = ops.conv2d_3x3s2(x, p.weight, p.bias)
            x = ops.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, self.training,
                                   act=ops.ACT_RELU)
        Wo, C = x.shape[2], x.shape[3]
        x = ops.rowscale_dropout(x.reshape(N, T, Wo * C), nonpad_rows)                            # masked_fill(mask, 0), :387-388
        g = self.gru
        # reference feature order is (c, w); ours is (w, c): permute W_ih's columns once per call (96 x 256, autograd un-permutes)
        w_ih = g.weight_ih_l0.view(-1, C, Wo).transpose(1, 2).reshape(-1, Wo * C)
        return ops.linear(x, w_ih, g.bias_ih_l0), g.weight_hh_l0, g.bias_hh_l0


class StyleEmbedAttention(nn.Module):
    def __init__(self, query_dim, key_dim, num_units):
        super().__init__()
        self.W_query = _Linear(query_dim, num_units, bias=False)
        self.W_key = _Linear(key_dim, num_units, bias=False)
        self.W_value = _Linear(key_dim, num_units, bias=False)


class STL(nn.Module):
    """Style token layer with ONE head (modules.py:453-534)."""

    def __init__(self, model_config):
        super().__init__()
        E = model_config["transformer"]["encoder_hidden"]
        self.embed = nn.Parameter(torch.empty(model_config["prosody_modeling"]["liu2021"]["token_num"], E))
        self.attention = StyleEmbedAttention(E // 2, E, E)
        self.key_dim = E

    def forward(self, query):
        """query [N,E/2] -> [N,E]"""
        a = self.attention
        keys_in = torch.tanh(self.embed)                                     # [tokens,E]  (8 K elements)
        q = ops.linear(query, a.W_query.weight)
        k = ops.linear(keys_in, a.W_key.weight)
        v = ops.linear(keys_in, a.W_value.weight)
        sc = ops.linear(q, k, alpha=1.0 / math.sqrt(self.key_dim))           # q k^T / sqrt(d_k)   [N,tokens]
        sc = ops.masked_softmax(sc.unsqueeze(0)).squeeze(0)
        return ops.bmm_nn(sc.unsqueeze(0), v.unsqueeze(0)).squeeze(0)


class UtteranceLevelProsodyEncoder(nn.Module):
    """modules.py:537-569"""

    def __init__(self, preprocess_config, model_config):
        super().__init__()
        cfg = model_config["prosody_modeling"]["liu2021"]
        E = model_config["transformer"]["encoder_hidden"]
        self.encoder = ReferenceEncoder(preprocess_config, model_config)
        self.encoder_prj = _Linear(cfg["ref_enc_gru_size"], E // 2)
        self.stl = STL(model_config)
        self.encoder_bottleneck = _Linear(E, cfg["bottleneck_size_u"])
        self.p_drop = cfg["ref_attention_dropout"]
        self.drop_ctx = None

    def forward(self, mel, nonpad_rows):
        return self.head(self.encoder(mel, nonpad_rows))

    def head(self, mem):
        last = mem[:, -1, :].contiguous()                                    # final hidden state after ALL padded steps
        ep = ops.linear(last, self.encoder_prj.weight, self.encoder_prj.bias)
        out = ops.linear(self.stl(ep), self.encoder_bottleneck.weight, self.encoder_bottleneck.bias)
        p = self.p_drop if self.training else 0.0
        if p > 0:
            out = ops.rowscale_dropout(out, None, p, self.drop_ctx)
        return out.unsqueeze(1)                                              # [N,1,bottleneck_u]


class _LinearNorm(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear = _Linear(cin, cout, bias=False)


class PhonemeLevelProsodyEncoder(nn.Module):
    """modules.py:400-450"""

    def __init__(self, preprocess_config, model_config):
        super().__init__()
        cfg = model_config["prosody_modeling"]["liu2021"]
        self.E = model_config["transformer"]["encoder_hidden"]
        self.encoder = ReferenceEncoder(preprocess_config, model_config)
        self.linears = nn.ModuleList([_LinearNorm(self.E, self.E), _LinearNorm(self.E, self.E)])
        self.encoder_prj = _Linear(cfg["ref_enc_gru_size"], 2 * self.E)
        self.encoder_bottleneck = _Linear(self.E, cfg["bottleneck_size_p"])
        self.p_drop = cfg["ref_attention_dropout"]
        self.drop_ctx = None

    def forward(self, x, src_len_i32, src_nonpad_rows, mel, mel_len_i32, mel_nonpad_rows):
        """x [N,Ts,E] -> (out [N,Ts,bottleneck_p], attn [N,Ts,Tm])"""
        return self.head(x, src_len_i32, src_nonpad_rows, mel_len_i32, self.encoder(mel, mel_nonpad_rows))

    def head(self, x, src_len_i32, src_nonpad_rows, mel_len_i32, mem):
        E = self.E
        ep = ops.linear(mem.contiguous(), self.encoder_prj.weight, self.encoder_prj.bias)         # [N,Tm,2E]
        k, v = ep[..., :E], ep[..., E:]
        q = ops.linear(x, self.linears[0].linear.weight)
        k = ops.linear(k.contiguous(), self.linears[1].linear.weight)
        attn = ops.bmm_nt(q, k, 1.0 / math.sqrt(E))                                               # [N,Ts,Tm]
        p = self.p_drop if self.training else 0.0
        if p > 0:       # dropout(softmax) THEN masked_fill(text_mask, 0): the zero rows commute with the dropout mask
            attn = ops.masked_softmax(attn, mel_len_i32, None)
            attn = ops.rowscale_dropout(attn, src_nonpad_rows, p, self.drop_ctx)
        else:
            attn = ops.masked_softmax(attn, mel_len_i32, src_len_i32)
        ctx = ops.bmm_nn(attn, v)
        out = ops.linear(ctx, self.encoder_bottleneck.weight, self.encoder_bottleneck.bias, rowscale=src_nonpad_rows)
        return out, attn


class ParallelProsodyPredictor(nn.Module):
    """modules.py:572-648: 2x [Conv1d k -> ReLU -> nn.LayerNorm (eps 1e-5) -> dropout] -> bi-GRU -> Linear.  No padding masks:
    the GRU (and the utterance-level final states) see the padded positions exactly as in the reference."""

    def __init__(self, model_config, phoneme_level=True):
        super().__init__()
        cfg = model_config["prosody_modeling"]["liu2021"]
        E = model_config["transformer"]["encoder_hidden"]
        k = cfg["predictor_kernel_size"]
        if k != 3:
            raise NotImplementedError("ParallelProsodyPredictor: conv1d_2 hard-codes padding=1 (modules.py:607), i.e. kernel 3")
        self.E, self.phoneme_level, self.dropout = E, phoneme_level, cfg["predictor_dropout"]
        self.conv_layer = nn.Module()
        self.conv_layer.add_module("conv1d_1", _ConvNormK(E, E, k))
        self.conv_layer.add_module("layer_norm_1", _Norm(E))
        self.conv_layer.add_module("conv1d_2", _ConvNormK(E, E, k))
        self.conv_layer.add_module("layer_norm_2", _Norm(E))
        self.gru = _GRUParams(E, E // 2, True)
        self.predictor_bottleneck = _Linear(E, cfg["bottleneck_size_p"] if phoneme_level else cfg["bottleneck_size_u"])
        self.drop_ctx = None

    def forward(self, x):
        p = self.dropout if self.training else 0.0
        drop = self.drop_ctx if p > 0 else None
        cl = self.conv_layer
        for cv, ln in ((cl.conv1d_1.conv, cl.layer_norm_1), (cl.conv1d_2.conv, cl.layer_norm_2)):
            x = ops.conv1d(x, cv.weight, cv.bias, act=ops.ACT_RELU)
            x = ops.layer_norm(x, ln.weight, ln.bias, 1e-5, p_drop=p, drop=drop)
        mem = self.gru.run(x)                                                 # [N,T,E] = forward | backward halves
        H = self.E // 2
        if self.phoneme_level:
            pv = mem
        else:                                                                 # final states: forward at t = T-1, backward at t = 0
            pv = torch.cat([mem[:, -1, :H], mem[:, 0, H:]], -1).unsqueeze(1)
        return ops.linear(pv.contiguous(), self.predictor_bottleneck.weight, self.predictor_bottleneck.bias)


def reset_prosody_parameters(module):
    """torch default initialisers of the reference layers: nn.Linear / nn.Conv2d / nn.Conv1d kaiming_uniform(a=sqrt 5) + fan-in bias,
    nn.GRU U(-1/sqrt(H), 1/sqrt(H)), LinearNorm xavier_uniform, STL.embed N(0, 0.5) (modules.py:464)."""
    for name, p in module.named_parameters():
        name = "." + name
        if name.endswith("stl.embed"):
            nn.init.normal_(p, mean=0, std=0.5)
        elif ".gru." in name:
            H = p.shape[0] // 3
            nn.init.uniform_(p, -1 / math.sqrt(H), 1 / math.sqrt(H))
        elif ".linears." in name:
            nn.init.xavier_uniform_(p)
        elif "layer_norm" in name or ".bns." in name:
            (nn.init.ones_ if name.endswith("weight") else nn.init.zeros_)(p)
        elif name.endswith("weight") and p.dim() >= 2:
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
        elif name.endswith("bias"):
            w = dict(module.named_parameters())[name[1:-4] + "weight"]
            fan_in = w[0].numel()
            nn.init.uniform_(p, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
