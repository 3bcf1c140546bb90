"""Drop-in `CompTransTTS` (reference: model/CompTransTTS.py:12-152) and the `transformer_fs2`
block plugin (`TextEncoder` / `Decoder`, reference: model/transformers/transformer_fs2.py)
re-built on the gfx950 kernels of libctts_hip.so.

Contract kept from the reference:
  * constructor `(preprocess_config, model_config, train_config)` and the positional
    `forward(speakers, texts, src_lens, max_src_len, mels, mel_lens, max_mel_len, p_targets,
    e_targets, d_targets, attn_priors, spker_embeds, p_control, e_control, d_control, step)`
    returning the same 14-tuple (CompTransTTS.py:64-82,137-152);
  * `state_dict()` key names / shapes so released checkpoints load;
  * the `block_type` plugin surface: `TextEncoder(config)`, `Decoder(config)`, `.d_model`.

Not a module-tree translation: activations stay [B,T,C]; each reference sub-layer maps to one
or two fused kernel launches (see ops.py).  Supported here (= every BASELINE.json config): block_type transformer_fs2 and
conformer (conformer.py); learn_alignment False and True (aligner + device MAS, single- and multi-speaker); prosody_modeling
"none" and "liu2021" (prosody.py); pitch_type cwt / frame / ph (pitch_norm log / standard, use_uv); phoneme- and frame-level
energy; ffn_act gelu / relu / swish ...; ffn_padding SAME / LEFT; use_pitch_embed / use_energy_embed on or off.  Values the reference
does not build either (prosody du2021 aside, which is outside SURVEY.md section 8; pitch_ar, which the reference cannot run) raise
NotImplementedError in the constructor - nothing is silently ignored.
"""
import json
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from . import ops
from .configs import N_SYMBOLS

F0_BIN = 256
F0_MEL_MIN = 1127 * math.log(1 + 50.0 / 700)
F0_MEL_MAX = 1127 * math.log(1 + 1100.0 / 700)


# --------------------------------------------------------------------------- parameter holders
class _Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None


class _Conv(nn.Module):
    """Conv1d parameters with the reference's shape [Cout, Cin, K] (state-dict / optimizer compatible) but GEMM-major MEMORY
    [Cout][K][Cin] behind permuted strides: the implicit-GEMM forward reads the weight as is, the weight-gradient GEMM accumulates
    straight into `.grad` (same strides), only the data-gradient operand is repacked (ops._LinearConv)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, k, cin).permute(0, 2, 1))
        self.bias = nn.Parameter(torch.empty(cout))


class _Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _BatchNorm(_Norm):
    def __init__(self, c):
        super().__init__(c)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _PosEmbed(nn.Module):
    """fs2 SinusoidalPositionalEmbedding (blocks.py:49-108): only the `_float_tensor` buffer is
    state; the table itself is regenerated (and grown on demand) on the module's device."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.register_buffer("_float_tensor", torch.zeros(1))
        self._table = None

    def table(self, n_pos):
        dev = self._float_tensor.device
        if self._table is None or self._table.shape[0] < n_pos or self._table.device != dev:
            self._table = ops.sinusoid_table(max(n_pos, 1025), self.dim, dev)
        return self._table

    def lookup(self, src, stride):
        """src: tokens [B,T] int64 or activations [B,T,C] float32 (channel 0 != 0 marks non-pad)."""
        pos = K.positions(src, stride)
        tab = self.table(src.shape[1] + 1)
        return F.embedding(pos.long(), tab)

    def add_to(self, x, src, stride, alpha=None, rowscale=None, p_drop=0.0, drop=None):
        """rowscale * dropout(x + alpha * embed_positions(src)) in one launch each way (ops.posembed_add); positions from `src`
        (tokens, or the activations themselves: blocks.py:85-104 make_positions on channel 0)."""
        pos = K.positions(src, stride)
        return ops.posembed_add(x, pos, self.table(src.shape[1] + 1), alpha, rowscale, p_drop, drop)


class _SelfAttn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * c, c))
        self.out_proj = _Linear(c, c, bias=False)


class _FFN(nn.Module):
    def __init__(self, c, k, padding="SAME"):
        super().__init__()
        # transformer_fs2.py:209-215: LEFT wraps the convolution as nn.Sequential(ConstantPad1d, Conv1d) - its parameters are then
        # called ffn_1.1.weight / ffn_1.1.bias in a state dict
        self.ffn_1 = _Conv(c, 4 * c, k) if padding == "SAME" else nn.ModuleDict({"1": _Conv(c, 4 * c, k)})
        self.ffn_2 = _Linear(4 * c, c)

    @property
    def conv1(self):
        return self.ffn_1["1"] if isinstance(self.ffn_1, nn.ModuleDict) else self.ffn_1


class _EncSALayer(nn.Module):
    def __init__(self, c, k, padding="SAME"):
        super().__init__()
        self.layer_norm1 = _Norm(c)
        self.self_attn = _SelfAttn(c)
        self.layer_norm2 = _Norm(c)
        self.ffn = _FFN(c, k, padding)


class _Layer(nn.Module):
    def __init__(self, c, k, padding="SAME"):
        super().__init__()
        self.op = _EncSALayer(c, k, padding)


def mask_aux(pad_mask, lens=None):
    """(non-pad row scale float32 [B*T], valid lengths int32 [B]) of a padding mask [B,T] (True = pad), computed ONCE per mask object:
    the encoder / decoder stack, the predictors and the variance adaptor all ask for the same two tensors (4 small launches each time)."""
    aux = getattr(pad_mask, "_ctts_aux", None)
    if aux is None:
        valid = ~pad_mask
        nonpad = valid.to(torch.float32).reshape(-1).contiguous()
        if lens is None and pad_mask.is_cuda:         # row sums of the mask through the ordered column-sum kernel (no torch reduction on a captured path)
            B_, T_ = pad_mask.shape
            lens = K.colsum(nonpad.view(B_, T_).t().contiguous())
        elif lens is None:                           # host tensors (collate-side helpers, CPU tests)
            lens = valid.sum(1)
        li = lens.to(torch.int32).contiguous()
        aux = (nonpad, li)
        try:
            pad_mask._ctts_aux = aux
        except Exception:        # noqa: BLE001   (a tensor subclass without a __dict__)
            pass
    return aux


def _runtime(module):
    """dropout context shared by the whole model (device-resident seed, see kernels.DropCtx)."""
    ctx = getattr(module, "_drop_ctx", None)
    dev = next(module.parameters()).device
    if ctx is None or ctx.seed.device != dev:
        ctx = K.DropCtx(dev)
        module._drop_ctx = ctx
    return ctx


class FFTBlocks(nn.Module):
    """reference: transformer_fs2.py:16-72 (FFTBlocks), :154-200 (EncSALayer), :203-239 (FFN)."""

    def __init__(self, hidden, n_layers, ksize, dropout, n_heads, use_pos_embed, ffn_act=ops.ACT_GELU, ffn_padding="SAME"):
        super().__init__()
        self.hidden_size, self.num_layers, self.ksize = hidden, n_layers, ksize
        self.dropout, self.num_heads, self.use_pos_embed = dropout, n_heads, use_pos_embed
        self.ffn_act, self.ffn_padding = ffn_act, ffn_padding
        if use_pos_embed:
            self.pos_embed_alpha = nn.Parameter(torch.ones(1))
            self.embed_positions = _PosEmbed(hidden)
        self.layers = nn.ModuleList([_Layer(hidden, ksize, ffn_padding) for _ in range(n_layers)])
        self.layer_norm = _Norm(hidden)
        self.drop_ctx = None  # set by the owning model
        self._cut_prefix = None   # "decoder.layers" when owned by CompTransTTS: names of the staged-backward cut points (dp.stage_plan)

    def run(self, x, pad_mask):
        """x [B,T,C] float32, pad_mask [B,T] bool (True = pad) -> [B,T,C]"""
        B, T, C = x.shape
        p = self.dropout if self.training else 0.0
        drop = self.drop_ctx if p > 0 else None
        nonpad, lens = mask_aux(pad_mask)
        if self.use_pos_embed:
            x = self.embed_positions.add_to(x.contiguous(), x, C, self.pos_embed_alpha, nonpad, p, drop)
        else:
            x = ops.rowscale_dropout(x, nonpad, 0.0, None)
        alpha = self.ksize ** -0.5
        pr = ops.PadRows(lens, T)      # rows t >= len are padding: every sub-layer output is re-masked (transformer_fs2.py:190,199)
        for li, layer in enumerate(self.layers):
            op = layer.op
            if self._cut_prefix is not None:
                x = ops.stage_cut(x, f"{self._cut_prefix}.{li}")
            h, xr = ops.layer_norm_res(x, op.layer_norm1.weight, op.layer_norm1.bias, 1e-12)
            qkv = ops.linear(h, op.self_attn.in_proj_weight, pad_rows=pr)
            a = ops.self_attention(qkv, lens, self.num_heads)
            x = ops.linear(a, op.self_attn.out_proj.weight, None, residual=xr, rowscale=nonpad, p_drop=p, drop=drop, pad_rows=pr)
            # the FFN convolution runs on the plane kernel at the decoder's size: LN2 writes its operand planes in the same launch
            h, xr = ops.layer_norm_res(x, op.layer_norm2.weight, op.layer_norm2.bias, 1e-12,
                                       planes_for=(op.ffn.conv1.weight.shape[0], self.ksize))
            # g feeds ffn_2 only: its epilogue backward (GELU', dropout mask) rides in the epilogue of ffn_2's data-gradient GEMM
            link = ops.EpiLink()
            g = ops.conv1d(h, op.ffn.conv1.weight, op.ffn.conv1.bias, act=self.ffn_act, alpha=alpha, p_drop=p, drop=drop,
                           pad_rows=pr, link=link, link_role=1, padding=self.ffn_padding)
            x = ops.linear(g, op.ffn.ffn_2.weight, op.ffn.ffn_2.bias, residual=xr, rowscale=nonpad, p_drop=p, drop=drop,
                           pad_rows=pr, link=link, link_role=2)
        return ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, 1e-5, rowscale=nonpad)

    def forward(self, x, padding_mask=None):
        if padding_mask is None:
            padding_mask = x.abs().sum(-1).eq(0)
        return self.run(x, padding_mask), padding_mask


_FFN_ACTS = {"gelu": ops.ACT_GELU, "relu": ops.ACT_RELU, "swish": ops.ACT_SWISH}


def _ffn_switches(config):
    """transformer_fs2.py:87-88,131-132 hand variance_predictor.ffn_padding / ffn_act to every FFN (TransformerFFNLayer :203-239: act in
    gelu / relu / swish - any other string means NO activation there -, padding SAME / LEFT) -> (activation code, padding)"""
    vp = config.get("variance_predictor", {})
    act, padding = vp.get("ffn_act", "gelu"), vp.get("ffn_padding", "SAME")
    if padding not in ("SAME", "LEFT"):
        raise NotImplementedError(f"variance_predictor.ffn_padding '{padding}': the reference builds SAME and LEFT only")
    return _FFN_ACTS.get(act, ops.ACT_NONE), padding


class TextEncoder(FFTBlocks):
    """plugin contract: TextEncoder(config).forward(tokens[B,Ts], pad_mask[B,Ts]) -> (enc, word_emb)."""

    def __init__(self, config):
        c = config["transformer_fs2"]
        act, padding = _ffn_switches(config)
        super().__init__(c["encoder_hidden"], c["encoder_layer"], c["ffn_kernel_size"], c["encoder_dropout"],
                         c["encoder_head"], use_pos_embed=False, ffn_act=act, ffn_padding=padding)
        self.embed_tokens = nn.Embedding(N_SYMBOLS + 1, c["encoder_hidden"], padding_idx=0)
        self.embed_positions = _PosEmbed(c["encoder_hidden"])
        self.embed_scale = math.sqrt(c["encoder_hidden"])
        self.d_model = c["encoder_hidden"]

    def forward(self, txt_tokens, encoder_padding_mask):
        emb = self.embed_scale * ops.embedding(txt_tokens, self.embed_tokens.weight, 0)
        p = self.dropout if self.training else 0.0
        x = self.embed_positions.add_to(emb, txt_tokens.contiguous(), 1, None, None, p, self.drop_ctx if p > 0 else None)
        return self.run(x, encoder_padding_mask), emb


class Decoder(FFTBlocks):
    """plugin contract: Decoder(config).forward(x[B,Tm,H], pad_mask[B,Tm]) -> (dec, mask)."""

    def __init__(self, config):
        c = config["transformer_fs2"]
        act, padding = _ffn_switches(config)
        super().__init__(c["decoder_hidden"], c["decoder_layer"], c["ffn_kernel_size"], c["decoder_dropout"],
                         c["decoder_head"], use_pos_embed=True, ffn_act=act, ffn_padding=padding)
        self.d_model = c["decoder_hidden"]


# --------------------------------------------------------------------------- variance adaptor
class _PredictorConvs(nn.Module):
    """ModuleList of {1: Conv1d, 3: LayerNorm} so that keys read conv.{i}.1.weight / conv.{i}.3.weight."""

    def __init__(self, idim, n_layers, n_chans, ksize, padding="SAME"):
        super().__init__()
        self.padding = padding        # modules.py:1270-1283,1328-1331: ConstantPad1d (k-1)//2 on both sides (SAME) or (k-1, 0) (LEFT)
        self.conv = nn.ModuleList([
            nn.ModuleDict({"1": _Conv(idim if i == 0 else n_chans, n_chans, ksize), "3": _Norm(n_chans)})
            for i in range(n_layers)])

    def run_convs(self, x, p, drop, nonpad):
        for blk in self.conv:
            x = ops.conv1d(x, blk["1"].weight, blk["1"].bias, act=ops.ACT_RELU, padding=self.padding)
            x = ops.layer_norm(x, blk["3"].weight, blk["3"].bias, 1e-12, rowscale=nonpad, p_drop=p, drop=drop)
        return x


class DurationPredictor(_PredictorConvs):
    """reference: modules.py:1252-1310"""

    def __init__(self, idim, n_layers, n_chans, ksize, dropout, padding="SAME"):
        super().__init__(idim, n_layers, n_chans, ksize, padding)
        self.linear = _Linear(n_chans, 1)
        self.dropout = dropout
        self.drop_ctx = None

    def forward(self, x, src_pad):
        p = self.dropout if self.training else 0.0
        nonpad = mask_aux(src_pad)[0]
        h = self.run_convs(x, p, self.drop_ctx if p > 0 else None, nonpad)
        return ops.linear(h, self.linear.weight, self.linear.bias, rowscale=nonpad).squeeze(-1)


class PitchPredictor(_PredictorConvs):
    """reference: modules.py:1313-1356 (also EnergyPredictor, :1359)"""

    def __init__(self, idim, n_layers, n_chans, odim, ksize, dropout, padding="SAME"):
        super().__init__(idim, n_layers, n_chans, ksize, padding)
        self.linear = _Linear(n_chans, odim)
        self.embed_positions = _PosEmbed(idim)
        self.pos_embed_alpha = nn.Parameter(torch.ones(1))
        self.dropout = dropout
        self.drop_ctx = None

    def forward(self, x, squeeze=False):
        p = self.dropout if self.training else 0.0
        x = x.contiguous()
        x = self.embed_positions.add_to(x, x, x.shape[-1], self.pos_embed_alpha)
        h = self.run_convs(x, p, self.drop_ctx if p > 0 else None, None)
        out = ops.linear(h, self.linear.weight, self.linear.bias)
        return out.squeeze(-1) if squeeze else out


class _CwtPredictor(nn.Module):
    """nn.Sequential(Linear, PitchPredictor) of modules.py:765-772 with keys '0' / '1'."""

    def __init__(self, hidden, h, filt, layers, odim, ksize, dropout, padding="SAME"):
        super().__init__()
        self.add_module("0", _Linear(hidden, h))
        self.add_module("1", PitchPredictor(h, layers, filt, odim, ksize, dropout, padding))

    def forward(self, x):
        lin, pp = getattr(self, "0"), getattr(self, "1")
        return pp(ops.linear(x, lin.weight, lin.bias))


class _StatsMLP(nn.Module):
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) of modules.py:773-776 (keys 0, 2, 4)."""

    def __init__(self, hidden, h):
        super().__init__()
        self.add_module("0", _Linear(hidden, h))
        self.add_module("2", _Linear(h, h))
        self.add_module("4", _Linear(h, 2))

    def forward(self, x):
        a, b, c = getattr(self, "0"), getattr(self, "2"), getattr(self, "4")
        x = ops.linear(x, a.weight, a.bias, act=ops.ACT_RELU)
        x = ops.linear(x, b.weight, b.bias, act=ops.ACT_RELU)
        return ops.linear(x, c.weight, c.bias)


def cwt2f0_norm(cwt_spec, mean, std, width, eps):
    """utils/pitch_tools.py:258-294 with pitch_norm == 'log' (tiny [B,Tm] elementwise math)."""
    b = (torch.arange(cwt_spec.shape[-1], dtype=torch.float32, device=cwt_spec.device) + 3.5) ** (-2.5)
    rec = (cwt_spec * b).sum(-1)
    rec = (rec - rec.mean(-1, keepdim=True)) / rec.std(-1, keepdim=True)
    f0 = (rec * std[:, None] + mean[:, None]).exp()
    if width > f0.shape[1]:
        f0 = torch.cat([f0] + [f0[:, -1:]] * (width - f0.shape[1]), 1)
    return torch.log2(f0 + eps)


def f0_to_coarse(f0):
    """utils/pitch_tools.py:27-36"""
    f0_mel = 1127 * (1 + f0 / 700).log()
    scaled = (f0_mel - F0_MEL_MIN) * (F0_BIN - 2) / (F0_MEL_MAX - F0_MEL_MIN) + 1
    f0_mel = torch.where(f0_mel > 0, scaled, f0_mel)
    f0_mel = f0_mel.clamp(min=1.0, max=float(F0_BIN - 1))
    return (f0_mel + 0.5).long()


class _ConvNormK(nn.Module):
    """ConvNorm wrapper (blocks.py:255-298): key `<idx>.conv.{weight,bias}`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = _Conv(cin, cout, k)


class AlignmentEncoder(nn.Module):
    """reference: modules.py:1117-1213 (single-speaker form).  key_proj: Conv k3 (256->512) ReLU Conv k1 (512->80);
    query_proj: Conv k3 (80->160) ReLU Conv k1 (160->80) ReLU Conv k1 (80->80); Gaussian isotropic scores."""

    def __init__(self, n_mel, n_att, n_text, temperature, multi_speaker):
        super().__init__()
        self.temperature = temperature
        self.key_proj = nn.Module()
        self.key_proj.add_module("0", _ConvNormK(n_text, 2 * n_text, 3))
        self.key_proj.add_module("2", _ConvNormK(2 * n_text, n_att, 1))
        self.query_proj = nn.Module()
        self.query_proj.add_module("0", _ConvNormK(n_mel, 2 * n_mel, 3))
        self.query_proj.add_module("2", _ConvNormK(2 * n_mel, n_mel, 1))
        self.query_proj.add_module("4", _ConvNormK(n_mel, n_att, 1))
        if multi_speaker:                     # modules.py:1172-1174 (bias-free LinearNorm)
            self.key_spk_proj = nn.Module()
            self.key_spk_proj.linear = _Linear(n_text, n_text, bias=False)
            self.query_spk_proj = nn.Module()
            self.query_spk_proj.linear = _Linear(n_text, n_mel, bias=False)

    def forward(self, mel, text_emb, src_pad, attn_prior, speaker_embedding=None):
        """mel [B,Tm,80], text_emb [B,Ts,256], src_pad [B,Ts] bool, attn_prior [B,Tm,Ts] -> (soft, logprob) [B,1,Tm,Ts]"""
        if speaker_embedding is not None:     # the projected speaker vector is added to every key / query position (:1188-1194)
            text_emb = ops.add_over_time(text_emb, ops.linear(speaker_embedding, self.key_spk_proj.linear.weight))
            mel = ops.add_over_time(mel, ops.linear(speaker_embedding, self.query_spk_proj.linear.weight))
        kp, qp = self.key_proj, self.query_proj
        k0, k2 = getattr(kp, "0").conv, getattr(kp, "2").conv
        q0, q2, q4 = getattr(qp, "0").conv, getattr(qp, "2").conv, getattr(qp, "4").conv
        k = ops.conv1d(text_emb, k0.weight, k0.bias, act=ops.ACT_RELU)
        k = ops.linear(k, k2.weight.view(k2.weight.shape[0], -1), k2.bias)
        q = ops.conv1d(mel, q0.weight, q0.bias, act=ops.ACT_RELU)
        q = ops.linear(q, q2.weight.view(q2.weight.shape[0], -1), q2.bias, act=ops.ACT_RELU)
        q = ops.linear(q, q4.weight.view(q4.weight.shape[0], -1), q4.bias)
        attn = ops.neg_sqdist(q, k, self.temperature)                                   # [B,Tm,Ts]
        attn = torch.log_softmax(attn, dim=-1) + torch.log(attn_prior + 1e-8)
        logprob = attn
        ops.mark_ready(logprob)               # the CTC forward-sum loss may start from here, beside the decoder (ops.mark_ready)
        soft = torch.softmax(attn.masked_fill(src_pad[:, None, :], float("-inf")), dim=-1)
        return soft.unsqueeze(1), logprob.unsqueeze(1)


def phoneme_level_mean(frame_values, dur, src_lens):
    """get_phoneme_level_energy (utils/tools.py:56-66, modules.py:882-888): per-phoneme mean of a frame-level
    feature over the phoneme's contiguous frame run, 0 where the duration is 0 - as a prefix-sum difference on device."""
    d = dur.long()
    ends = torch.cumsum(d, 1)
    starts = ends - d
    Tm = frame_values.shape[1]
    cs = torch.cat([frame_values.new_zeros(frame_values.shape[0], 1, dtype=torch.float64), frame_values.double().cumsum(1)], 1)
    seg = cs.gather(1, ends.clamp(max=Tm)) - cs.gather(1, starts.clamp(max=Tm))
    out = torch.where(d > 0, seg / d.clamp(min=1).double(), torch.zeros_like(seg)).float()
    valid = torch.arange(d.shape[1], device=d.device)[None, :] < src_lens[:, None]
    return out * valid


def phoneme_level_pitch(f0_frame, mel2ph, mel_lens, n_phones):
    """get_phoneme_level_pitch (utils/tools.py:47-53, modules.py:873-880): mean of the frame-level f0 over the frames whose mel2ph
    points at the phoneme (1-based; only the first mel_len frames count), 0-frame phonemes get 0 - as a one-hot contraction in fp64
    on the device (the reference scatter_adds per utterance on the host)."""
    B, Tm = f0_frame.shape
    dev = f0_frame.device
    valid = torch.arange(Tm, device=dev)[None, :] < mel_lens[:, None]
    # mel2ph is non-decreasing over an utterance's frames (the length regulator's contiguous runs): the frames of phoneme p are
    # [first index with mel2ph >= p, first index with mel2ph > p) - two binary searches per phoneme and a prefix-sum difference in fp64
    # instead of a [B, Tm, Ts] one-hot contraction (whose .sum(1) was a stock-torch reduction on the captured path)
    keys = torch.where(valid, mel2ph[:, :Tm].long(), torch.full_like(mel2ph[:, :Tm].long(), n_phones + 1)).contiguous()
    ids = torch.arange(1, n_phones + 1, device=dev)[None, :].expand(B, n_phones).contiguous()
    lo = torch.searchsorted(keys, ids, right=False)
    hi = torch.searchsorted(keys, ids, right=True)
    cs = torch.cat([f0_frame.new_zeros(B, 1, dtype=torch.float64), f0_frame.double().cumsum(1)], 1)
    tot = cs.gather(1, hi) - cs.gather(1, lo)
    cnt = (hi - lo).clamp(min=1).double()
    return (tot / cnt).float()


def denorm_f0(f0, uv, pitch_cfg, pitch_padding=None):
    """utils/pitch_tools.py:69-82 without the in-place writes (masks applied with torch.where)."""
    if pitch_cfg["pitch_norm"] == "standard":
        f0 = f0 * pitch_cfg["f0_std"] + pitch_cfg["f0_mean"]
    if pitch_cfg["pitch_norm"] == "log":
        f0 = 2 ** f0
    if uv is not None and pitch_cfg["use_uv"]:
        f0 = torch.where(uv > 0, torch.zeros_like(f0), f0)
    if pitch_padding is not None:
        f0 = torch.where(pitch_padding, torch.zeros_like(f0), f0)
    return f0


class VarianceAdaptor(nn.Module):
    """reference: modules.py:726-1114 (supervised, unsupervised (aligner + MAS) and inference branches)."""

    def __init__(self, preprocess_config, model_config, train_config, d_model):
        super().__init__()
        self.learn_alignment = model_config["duration_modeling"]["learn_alignment"]
        self.binarization_start_steps = train_config["duration"]["binarization_start_steps"]
        self.model_type = model_config["prosody_modeling"]["model_type"]
        if self.model_type not in ("none", "liu2021"):
            raise NotImplementedError(f"prosody_modeling.model_type '{self.model_type}': only 'none' and 'liu2021' are built "
                                      "(du2021 is outside SURVEY.md section 8)")
        pitch = preprocess_config["preprocessing"]["pitch"]
        self.pitch_type = pitch["pitch_type"]
        if self.pitch_type not in ("cwt", "frame", "ph"):
            raise NotImplementedError(f"pitch_type '{self.pitch_type}': the reference builds cwt, frame and ph (modules.py:754-786)")
        if pitch["pitch_norm"] not in ("log", "standard"):
            raise NotImplementedError(f"pitch_norm '{pitch['pitch_norm']}': the reference knows log and standard (pitch_tools.py:39-48)")
        if self.pitch_type == "cwt" and (pitch["pitch_norm"] != "log" or not pitch["use_uv"]):
            raise NotImplementedError("pitch_type=cwt is built for pitch_norm=log / use_uv=True (the shipped configs)")
        if pitch.get("pitch_ar", False):
            raise NotImplementedError("pitch_ar=True: the reference calls PitchPredictor(decoder_inp, f0) (modules.py:922), which its "
                                      "own PitchPredictor.forward(xs, squeeze) does not accept - nothing to reproduce")
        self.use_uv = bool(pitch["use_uv"])
        self.energy_level = preprocess_config["preprocessing"]["energy"]["feature"]       # utils/tools.py:30-44 get_variance_level
        if self.energy_level not in ("phoneme_level", "frame_level"):
            raise NotImplementedError(f"energy feature '{self.energy_level}': phoneme_level or frame_level")
        self.pitch_cfg = pitch
        vp = model_config["variance_predictor"]
        ve = model_config["variance_embedding"]
        # modules.py:735-736,754-821: the pitch / energy branches (predictors, embeddings, energy_bins) only exist when switched on
        self.use_pitch_embed, self.use_energy_embed = bool(ve.get("use_pitch_embed", True)), bool(ve.get("use_energy_embed", True))
        pad_mode = vp.get("ffn_padding", "SAME")          # modules.py:743: the predictors' ConstantPad1d follows it too
        if pad_mode not in ("SAME", "LEFT"):
            raise NotImplementedError(f"variance_predictor.ffn_padding '{pad_mode}': the reference builds SAME and LEFT only")
        self.predictor_grad = vp["predictor_grad"]
        self.cwt_std_scale = vp["cwt_std_scale"]
        hidden = model_config["transformer"]["encoder_hidden"]  # sic: modules.py:739 reads the 'transformer' section
        filt, drop = vp["filter_size"], vp["dropout"]
        # modules.py:788-799: learn_alignment reads the frame-level "unsup" statistics
        level_tag = "phone" if (not self.learn_alignment and self.energy_level == "phoneme_level") else "frame"
        stats_key = f"energy_{'unsup' if self.learn_alignment else 'sup'}_{level_tag}"
        n_ebins = model_config["variance_embedding"]["energy_n_bins"]
        if self.use_energy_embed:          # modules.py:788-799 reads stats.json inside this branch only: a stats file without the key is fine otherwise
            with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "stats.json")) as f:
                emin, emax = json.load(f)[stats_key][:2]
            if model_config["variance_embedding"]["energy_quantization"] == "log":
                bins = torch.exp(torch.linspace(math.log(emin), math.log(emax), n_ebins - 1))
            else:
                bins = torch.linspace(emin, emax, n_ebins - 1)
            self.energy_bins = nn.Parameter(bins, requires_grad=False)
        self.duration_predictor = DurationPredictor(hidden, vp["dur_predictor_layers"], filt, vp["dur_predictor_kernel"], drop, pad_mode)
        if self.use_pitch_embed and self.pitch_type == "cwt":
            self.cwt_predictor = _CwtPredictor(hidden, vp["cwt_hidden_size"], filt, vp["predictor_layers"], 11,
                                               vp["predictor_kernel"], drop, pad_mode)
            self.cwt_stats_layers = _StatsMLP(hidden, vp["cwt_hidden_size"])
        elif self.use_pitch_embed:          # modules.py:777-785: frame -> (f0, uv logit) per frame, ph -> f0 per phoneme
            self.pitch_predictor = PitchPredictor(hidden, vp["predictor_layers"], filt, 2 if self.pitch_type == "frame" else 1,
                                                  vp["predictor_kernel"], drop, pad_mode)
        if self.use_pitch_embed:
            self.pitch_embed = nn.Embedding(model_config["variance_embedding"]["pitch_n_bins"], hidden, padding_idx=0)
        if self.use_energy_embed:
            self.energy_predictor = PitchPredictor(hidden, vp["predictor_layers"], filt, 1, vp["predictor_kernel"], drop, pad_mode)
            self.energy_embedding = nn.Embedding(n_ebins, hidden, padding_idx=0)
        if self.learn_alignment:
            n_mel = preprocess_config["preprocessing"]["mel"]["n_mel_channels"]
            self.aligner = AlignmentEncoder(n_mel, n_mel, d_model, model_config["duration_modeling"]["aligner_temperature"],
                                            model_config["multi_speaker"])
        if self.model_type == "liu2021":          # modules.py:845-861
            from . import prosody as P
            cfg = model_config["prosody_modeling"]["liu2021"]
            self.utterance_prosody_encoder = P.UtteranceLevelProsodyEncoder(preprocess_config, model_config)
            self.phoneme_prosody_encoder = P.PhonemeLevelProsodyEncoder(preprocess_config, model_config)
            self.utterance_prosody_predictor = P.ParallelProsodyPredictor(model_config, phoneme_level=False)
            self.phoneme_prosody_predictor = P.ParallelProsodyPredictor(model_config, phoneme_level=True)
            self.utterance_prosody_prj = _Linear(cfg["bottleneck_size_u"], hidden)
            self.phoneme_prosody_prj = _Linear(cfg["bottleneck_size_p"], hidden)

    def _reference_memories(self, mel, mel_mask):
        """the part of the liu2021 prosody encoders that reads nothing but the target mel (modules.py:332-397,537-569): both reference
        encoders (Conv2d stack + BatchNorm, one launch for both 1,000-step GRUs) and the utterance-level head -> (up_emb, mem_p)"""
        mel_nonpad = mask_aux(mel_mask)[0]
        ue, pe = self.utterance_prosody_encoder, self.phoneme_prosody_encoder
        gi_u, whh_u, bhh_u = ue.encoder.features(mel, mel_nonpad)
        gi_p, whh_p, bhh_p = pe.encoder.features(mel, mel_nonpad)
        mem_u, mem_p = ops.gru_group([gi_u, gi_p], [whh_u, whh_p], [bhh_u, bhh_p])    # both Tm-step recurrences in one launch
        return ue.head(mem_u), mem_p

    def start_reference_encoders(self, mel, mel_mask):
        """called by CompTransTTS.forward BEFORE the text encoder: inside a train step (ops.side_loss_scope) the input-only branch runs
        on a side stream beside the text encoder; `_liu2021` joins.  Outside such a scope nothing happens here."""
        self._ref_pending = None
        if self.model_type != "liu2021" or not self.training or mel is None or mel_mask is None:
            return
        if not mel.is_cuda:
            return
        mask_aux(mel_mask)                                    # on the main stream, BEFORE the fork: the branch reads the cached tensors
        side = ops.fork_side(mel)
        if side is None:
            return
        with torch.cuda.stream(side):
            out = self._reference_memories(mel, mel_mask)
        self._ref_pending = (side, out)

    def _liu2021(self, x, src_len, src_mask, mel, mel_len, mel_mask):
        """Implicit prosody modelling branch (modules.py:1002-1022): encoders only in training, predictors always."""
        up_emb = pp_emb = pp_attn = None
        if self.training:
            assert mel is not None and mel_mask is not None, "liu2021 prosody encoders need the reference mel in training"
            src_nonpad = mask_aux(src_mask)[0]
            pend, self._ref_pending = getattr(self, "_ref_pending", None), None
            if pend is not None:
                side, (up_emb, mem_p) = pend
                ops.join_side(side)
                cur = torch.cuda.current_stream()
                up_emb.record_stream(cur)
                mem_p.record_stream(cur)
            else:
                up_emb, mem_p = self._reference_memories(mel, mel_mask)
            pp_emb, pp_attn = self.phoneme_prosody_encoder.head(x, src_len.to(torch.int32), src_nonpad, mel_len.to(torch.int32), mem_p)
        up_vec = self.utterance_prosody_predictor(x)
        u = up_emb if self.training else up_vec
        x = ops.add_over_time(x, ops.linear(u, self.utterance_prosody_prj.weight, self.utterance_prosody_prj.bias))      # [N,1,H] broadcast over Ts
        pp_vec = self.phoneme_prosody_predictor(x)
        pp = pp_emb if self.training else pp_vec
        x = ops.linear(pp, self.phoneme_prosody_prj.weight, self.phoneme_prosody_prj.bias, residual=x)
        return x, (up_emb, pp_emb, up_vec, pp_vec, pp_attn)

    def forward(self, speaker_embedding, text, text_embedding, src_len, src_mask, mel, mel_len, mel_mask=None,
                max_len=None, pitch_target=None, energy_target=None, duration_target=None, attn_prior=None,
                p_control=1.0, e_control=1.0, d_control=1.0, step=None):
        x = text
        if speaker_embedding is not None:
            x = ops.add_over_time(x, speaker_embedding)
        prosody_info = None
        if self.model_type == "liu2021":
            x, prosody_info = self._liu2021(x, src_len, src_mask, mel, mel_len, mel_mask)
        log_d = self.duration_predictor(ops.grad_scale(x, self.predictor_grad), src_mask)
        x_org = x
        attn_out = (None, None, None, None)
        if attn_prior is not None:      # training of unsupervised duration modelling (modules.py:1031-1053)
            assert self.learn_alignment and duration_target is None and mel is not None
            attn_soft, attn_logprob = self.aligner(mel, text_embedding, src_mask, attn_prior.transpose(1, 2), speaker_embedding)
            attn_hard, attn_hard_dur = ops.mas_binarize(attn_soft, src_len, mel_len)
            attn_out = (attn_soft, attn_hard, attn_hard_dur, attn_logprob)
            if step < self.binarization_start_steps:
                x = ops.bmm_nn(attn_soft.squeeze(1), x_org)
            else:
                x, mel_len, _ = ops.length_regulate(x_org, attn_hard_dur, max_len)
            d_rounded = attn_hard_dur
            mel2ph, _, _ = K.lr_index(d_rounded, int(max_len), pad=src_mask, round_mode=1)    # dur_to_mel2ph(...)[:, :max_len]
            pitch_target["mel2ph"] = mel2ph.long()
            if self.use_energy_embed and self.energy_level == "phoneme_level":        # modules.py:1095-1097
                energy_target = phoneme_level_mean(energy_target, attn_hard_dur, src_len)
            mel2ph = None
        elif duration_target is not None:
            assert not self.learn_alignment
            x, mel_len, _ = ops.length_regulate(x_org, duration_target, max_len)
            d_rounded = duration_target
            mel2ph = None
        else:
            d_rounded = torch.clamp(torch.round(torch.exp(log_d.detach()) - 1) * d_control, min=0)
            x, mel_len, _ = ops.length_regulate(x_org, d_rounded, max_len)
            ids = torch.arange(x.shape[1], device=x.device)[None, :]
            mel_mask = ids >= mel_len[:, None]
            mel2ph = ops.dur_to_mel2ph(d_rounded, src_mask)
        pitch_prediction = energy_prediction = None          # modules.py:982: stay None when the branch is switched off
        out = x
        if self.use_pitch_embed and self.pitch_type == "cwt":
            # ---- pitch (cwt)   modules.py:890-948,1071-1091
            cwt = self.cwt_predictor(ops.grad_scale(x, self.predictor_grad)) * p_control
            stats = self.cwt_stats_layers(x_org[:, 0, :].contiguous())
            f0_mean, f0_std = stats[:, 0], stats[:, 1]
            eps = self.pitch_cfg["pitch_norm_eps"]
            with torch.no_grad():
                # inverse CWT -> normalisation -> f0 -> 2 ** f0 with the uv mask -> f0_to_coarse: ONE launch (csrc/pitch.hip) instead of
                # ~25 stock-torch ones incl. the multi-block mean / std reductions (cwt2f0_norm / f0_to_coarse above are the same
                # arithmetic op by op: the CPU tests and tests/test_kernels_gpu.py compare the kernel with them)
                pk = dict(eps=eps, mel_min=F0_MEL_MIN, mel_max=F0_MEL_MAX, f0_bin=F0_BIN)
                if pitch_target is not None:
                    mel2ph = pitch_target["mel2ph"]
                    f0, f0_denorm, pitch_ids = K.cwt_pitch(pitch_target["cwt_spec"], pitch_target["f0_mean"], pitch_target["f0_std"], 1.0,
                                                           uv=pitch_target["uv"], width=mel2ph.shape[1], **pk)
                    pitch_target["f0"] = f0
                    pitch_target.update({"f0_cwt": pitch_target["f0"]})
                else:
                    w = mel2ph.shape[1]
                    if w == cwt.shape[1]:
                        f0, f0_denorm, pitch_ids = K.cwt_pitch(cwt, f0_mean, f0_std, self.cwt_std_scale, uv_chan=cwt.shape[-1] - 1,
                                                               nscale=cwt.shape[-1] - 1, **pk)
                    else:                    # the reference repeats the last f0 column up to mel2ph's width; uv is then indexed at that width too
                        uvf = (cwt[:, :, -1] > 0).float()
                        uvf = torch.cat([uvf] + [uvf[:, -1:]] * (w - uvf.shape[1]), 1).contiguous()
                        f0, f0_denorm, pitch_ids = K.cwt_pitch(cwt, f0_mean, f0_std, self.cwt_std_scale, uv=uvf,
                                                               nscale=cwt.shape[-1] - 1, width=w, **pk)
            pitch_embedding = ops.embedding(pitch_ids, self.pitch_embed.weight, 0)
            pitch_prediction = {"pitch_pred": None, "f0_denorm": f0_denorm, "cwt": cwt, "f0_mean": f0_mean, "f0_std": f0_std}
            out = out + pitch_embedding
        elif self.use_pitch_embed and self.pitch_type == "frame":
            # ---- pitch (frame)   modules.py:906-938: [f0, uv logit] per frame from the regulated sequence
            pitch_pred = self.pitch_predictor(ops.grad_scale(x, self.predictor_grad)) * p_control
            with torch.no_grad():
                if pitch_target is not None:
                    mel2ph = pitch_target["mel2ph"]
                    f0, uv = pitch_target["f0"], pitch_target["uv"]
                else:
                    f0 = pitch_pred[:, :, 0]
                    uv = (pitch_pred[:, :, 1] > 0) if self.use_uv else None
                pad = mel2ph[:, : f0.shape[1]] == 0
                f0_denorm = denorm_f0(f0, uv, self.pitch_cfg, pitch_padding=pad)
                if pitch_target is not None:
                    pitch_target["f0"] = torch.where(pad, torch.zeros_like(f0), f0)       # modules.py:934-935 (in place there)
                else:
                    pitch_pred[:, :, 0].masked_fill_(pad, 0.0)        # ... where f0 is a VIEW of the prediction (inference output)
                pitch_ids = f0_to_coarse(f0_denorm)
            pitch_embedding = ops.embedding(pitch_ids, self.pitch_embed.weight, 0)
            pitch_prediction = {"pitch_pred": pitch_pred, "f0_denorm": f0_denorm, "cwt": None, "f0_mean": None, "f0_std": None}
            out = out + pitch_embedding
        elif self.use_pitch_embed:
            # ---- pitch (ph)   modules.py:892-905: one f0 per phoneme from the encoder side, gathered to frames through mel2ph
            pitch_pred = self.pitch_predictor(ops.grad_scale(x_org, self.predictor_grad)) * p_control
            with torch.no_grad():
                if pitch_target is not None:
                    mel2ph = pitch_target["mel2ph"]
                    # modules.py:1083-1084 replaces the dict's frame-level contour by the phoneme-level one; a TrainStep calls forward
                    # again with the SAME dict (static graph inputs), so the frame-level tensor is kept under its own key
                    frame_f0 = pitch_target.setdefault("f0_frame", pitch_target["f0"])
                    pitch_target["f0"] = phoneme_level_pitch(frame_f0, mel2ph, mel_len, x_org.shape[1])
                    f0 = pitch_target["f0"]
                else:
                    f0 = pitch_pred[:, :, 0]
                # modules.py:895 `encoder_out.sum().abs() == 0` (a flag, never a sync here) through the ordered sum kernel: torch's
                # multi-block reduction over 0.5 M elements is the kind of launch that mis-replays under hipGraph
                all_pad = ops.sum_all(x_org).abs().reshape(()) == 0
                f0_denorm = denorm_f0(f0, None, self.pitch_cfg, pitch_padding=all_pad)
                ph_ids = F.pad(f0_to_coarse(f0_denorm), [1, 0])
                pitch_ids = torch.gather(ph_ids, 1, mel2ph.long())
            pitch_embedding = ops.embedding(pitch_ids, self.pitch_embed.weight, 0)
            pitch_prediction = {"pitch_pred": pitch_pred, "f0_denorm": f0_denorm, "cwt": None, "f0_mean": None, "f0_std": None}
            out = out + pitch_embedding
        if self.use_energy_embed:
            # ---- energy   modules.py:950-960,1092-1099 (no gradient scaling: :951 is a no-op); frame level reads the regulated sequence
            frame_e = self.energy_level == "frame_level"
            energy_prediction = self.energy_predictor(x if frame_e else x_org, squeeze=True)
            if energy_target is not None:
                e_ids = torch.bucketize(energy_target, self.energy_bins)
            else:
                energy_prediction = energy_prediction * e_control
                e_ids = torch.bucketize(energy_prediction.detach(), self.energy_bins)
            energy_embedding = ops.embedding(e_ids, self.energy_embedding.weight, 0)
            if frame_e:
                out = out + energy_embedding
            else:
                e_frames, _, _ = ops.length_regulate(energy_embedding, d_rounded, max_len)
                out = out + e_frames
        x = out
        return (x, pitch_target, pitch_prediction, energy_target, energy_prediction, log_d, d_rounded, mel_len, mel_mask,
                attn_out, prosody_info)


# --------------------------------------------------------------------------- postnet
class _ConvNorm(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = _Conv(cin, cout, k)


class PostNet(nn.Module):
    """reference: modules.py:78-148.  5 x (Conv1d k5 -> BatchNorm1d -> tanh (not last) -> dropout 0.5)."""

    def __init__(self, n_mel=80, dim=512, k=5, n=5):
        super().__init__()
        chans = [n_mel] + [dim] * (n - 1) + [n_mel]
        self.convolutions = nn.ModuleList()
        for i in range(n):
            seq = nn.Module()
            seq.add_module("0", _ConvNorm(chans[i], chans[i + 1], k))
            seq.add_module("1", _BatchNorm(chans[i + 1]))
            self.convolutions.append(seq)
        self.dropout = 0.5   # hard-coded in the reference (modules.py:144-145)
        self.drop_ctx = None

    def forward(self, x):
        n = len(self.convolutions)
        p = self.dropout if self.training else 0.0
        drop = self.drop_ctx if p > 0 else None
        convs = [getattr(seq, "0").conv for seq in self.convolutions]
        for i, seq in enumerate(self.convolutions):
            cv, bn = convs[i], getattr(seq, "1")
            cin_i, k_i = cv.weight.shape[1], cv.weight.shape[2]
            x = ops.conv1d(x, cv.weight, cv.bias)
            # operand planes from the producers (round 6): this layer's output is the A operand of the NEXT convolution's forward, and the
            # gradient BatchNorm's backward returns is the dZ operand of THIS convolution's data / weight gradient - when those launches
            # run on the plane kernel, the BatchNorm launches write the bf16 plane sets next to the fp32 tensors
            fwd_pl = i + 1 < n and ops.consumer_takes_planes(x, convs[i + 1].weight.shape[0], convs[i + 1].weight.shape[2])
            dz_pl = (self.training and i > 0 and x.is_cuda
                     and ops.consumer_takes_planes(x, cin_i, k_i))          # data gradient: [M, cout] x [cin, k * cout]^T
            x = ops.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                   self.training, act=ops.ACT_TANH if i < n - 1 else ops.ACT_NONE, p_drop=p, drop=drop,
                                   planes=fwd_pl, dx_planes=dz_pl)
        return x


# --------------------------------------------------------------------------- facade
class CompTransTTS(nn.Module):
    """Drop-in for model/CompTransTTS.py:12-152."""

    def __init__(self, preprocess_config, model_config, train_config):
        super().__init__()
        self.model_config = model_config
        bt = model_config["block_type"]
        if bt == "transformer_fs2":
            enc_cls, dec_cls = TextEncoder, Decoder
        elif bt == "conformer":
            from .conformer import TextEncoder as enc_cls, Decoder as dec_cls
        elif bt in ("transformer", "lstransformer", "fastformer", "reformer"):
            raise NotImplementedError(f"block_type '{bt}' has no MI355X-native plugin (SURVEY.md section 2 #6: out of scope; the "
                                      "plugin contract is kept, so a third-party TextEncoder/Decoder pair still plugs in)")
        else:
            raise NotImplementedError
        self.encoder = enc_cls(model_config)
        self.variance_adaptor = VarianceAdaptor(preprocess_config, model_config, train_config, self.encoder.d_model)
        self.decoder = dec_cls(model_config)
        self.decoder._cut_prefix = "decoder.layers" if hasattr(self.decoder, "layers") else "decoder.layer_stack"
        self.mel_linear = _Linear(self.decoder.d_model, preprocess_config["preprocessing"]["mel"]["n_mel_channels"])
        self.postnet = PostNet()
        self.speaker_emb = None
        if model_config["multi_speaker"]:
            self.embedder_type = preprocess_config["preprocessing"]["speaker_embedder"]
            if self.embedder_type == "none":
                with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "speakers.json")) as f:
                    n_speaker = len(json.load(f))
                self.speaker_emb = nn.Embedding(n_speaker, self.encoder.d_model)
            else:
                self.speaker_emb = _Linear(model_config["external_speaker_dim"], self.encoder.d_model)
        self.reset_parameters()

    # reference initialisers restated (blocks.py:10-23 Embedding/Linear, transformer_fs2.py:328-344 attention,
    # blocks.py:286-288 ConvNorm; torch defaults for nn.Conv1d / nn.Linear elsewhere)
    def reset_parameters(self):
        params = dict(self.named_parameters())
        conformer = self.model_config["block_type"] == "conformer"
        if conformer:
            from .conformer import reset_conformer_parameters
            reset_conformer_parameters(self.encoder)
            reset_conformer_parameters(self.decoder)
        prosody_mods = [m for n, m in self.variance_adaptor.named_children() if "prosody" in n and not n.endswith("_prj")]
        if prosody_mods:
            from .prosody import reset_prosody_parameters
            for m in prosody_mods:
                reset_prosody_parameters(m)
        for name, p in params.items():
            if name.endswith("pos_embed_alpha") or name.endswith("energy_bins"):
                continue
            if name.startswith("variance_adaptor.") and "_prosody_" in name and "_prosody_prj." not in name:
                continue
            if conformer and (name.startswith("encoder.") or name.startswith("decoder.")):
                continue
            if name.endswith("in_proj_weight") or name.endswith("out_proj.weight") or name.endswith("ffn_2.weight"):
                nn.init.xavier_uniform_(p)
            elif name.endswith("ffn_2.bias"):
                nn.init.zeros_(p)
            elif name.startswith("variance_adaptor.aligner.") and name.endswith("_spk_proj.linear.weight"):
                nn.init.xavier_uniform_(p)
            elif name.startswith("variance_adaptor.aligner.") and name.endswith("conv.weight"):
                relu = name.endswith("key_proj.0.conv.weight") or name.endswith("query_proj.0.conv.weight")
                nn.init.xavier_uniform_(p, gain=nn.init.calculate_gain("relu" if relu else "linear"))
            elif name.startswith("postnet.") and name.endswith("conv.weight"):
                last = name.startswith(f"postnet.convolutions.{len(self.postnet.convolutions) - 1}.")
                nn.init.xavier_uniform_(p, gain=nn.init.calculate_gain("linear" if last else "tanh"))
            elif name in ("encoder.embed_tokens.weight", "variance_adaptor.pitch_embed.weight",
                          "variance_adaptor.energy_embedding.weight"):
                nn.init.normal_(p, mean=0, std=p.shape[1] ** -0.5)
                with torch.no_grad():
                    p[0].zero_()
            elif name == "speaker_emb.weight" and p.dim() == 2 and isinstance(self.speaker_emb, nn.Embedding):
                nn.init.normal_(p)
            elif p.dim() >= 2:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            elif name.endswith(".bias") and (name[:-4] + "weight") in params and params[name[:-4] + "weight"].dim() >= 2:
                fan_in = params[name[:-4] + "weight"][0].numel()
                bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                nn.init.uniform_(p, -bound, bound)
            # 1-D norm weights / biases keep their ones / zeros

    def _wire(self):
        ctx = _runtime(self)
        for m in self.modules():
            if hasattr(m, "drop_ctx"):
                m.drop_ctx = ctx
        return ctx

    def forward(self, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None, max_mel_len=None,
                p_targets=None, e_targets=None, d_targets=None, attn_priors=None, spker_embeds=None,
                p_control=1.0, e_control=1.0, d_control=1.0, step=None):
        ctx = self._wire()
        ctx.begin_step()
        if self.training:
            ctx.advance()
        dev = texts.device
        src_masks = torch.arange(max_src_len, device=dev)[None, :] >= src_lens[:, None]
        mel_masks = (torch.arange(max_mel_len, device=dev)[None, :] >= mel_lens[:, None]) if mel_lens is not None else None
        mask_aux(src_masks, src_lens.clamp(max=max_src_len))          # the lengths are known here: no row sums of the masks later
        if mel_masks is not None:
            mask_aux(mel_masks, mel_lens.clamp(max=max_mel_len))
        if mels is not None and mel_masks is not None:
            self.variance_adaptor.start_reference_encoders(mels, mel_masks)      # liu2021, inside a train step: beside the text encoder
        enc, text_embeds = self.encoder(texts, src_masks)
        speaker_embeds = None
        if self.speaker_emb is not None:
            if self.embedder_type == "none":
                speaker_embeds = self.speaker_emb(speakers)
            else:
                assert spker_embeds is not None, "Speaker embedding should not be None"
                speaker_embeds = ops.linear(spker_embeds, self.speaker_emb.weight, self.speaker_emb.bias)
        va_out = self.variance_adaptor(
            speaker_embeds, enc, text_embeds, src_lens, src_masks, mels, mel_lens, mel_masks, max_mel_len, p_targets,
            e_targets, d_targets, attn_priors, p_control, e_control, d_control, step)
        # staged backward (dp.py): everything that leaves the variance adaptor - the decoder input and the predictions the loss
        # reads - is one cut, so the encoder + adaptor region is back-propagated once, after the decoder stages
        (output, p_targets, p_predictions, e_targets, e_predictions, log_d_predictions, d_rounded, mel_lens, mel_masks,
         attn_outs, prosody_info) = ops.stage_cut_tree(va_out, "decoder.in")
        output, mel_masks = self.decoder(output, mel_masks)
        output = ops.linear(output, self.mel_linear.weight, self.mel_linear.bias)
        postnet_output = self.postnet(output) + output
        return (output, postnet_output, p_predictions, e_predictions, log_d_predictions, d_rounded, src_masks, mel_masks,
                src_lens, mel_lens, attn_outs, prosody_info, p_targets, e_targets)
