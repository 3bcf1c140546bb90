"""TEST INFRASTRUCTURE - CPU restatement (oracle) of the reference hot path.

This module is the *checker*, never the product: only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  The product package
(comprehensive-transformer-tts_amd/) must never import anything under oracle/.

It restates, as plain functional torch-CPU fp32 code driven by a reference-keyed
state dict, what the reference computes on the path SURVEY.md section 8(a) lists:

  a1  CompTransTTS.forward                 model/CompTransTTS.py:64-152
  a2  get_mask_from_lengths                utils/tools.py:188-196
  a3  embedding + sinusoid positions       model/transformers/transformer_fs2.py:113-119,
                                           model/transformers/blocks.py:49-108, utils/tools.py:640-652
  a4  FFTBlocks.forward                    model/transformers/transformer_fs2.py:47-72
  a5  EncSALayer / MultiheadAttention      model/transformers/transformer_fs2.py:176-200,348-394
  a6  TransformerFFNLayer                  model/transformers/transformer_fs2.py:220-239
  a7  DurationPredictor                    model/modules.py:1299-1310
  a8  LengthRegulator + pad                model/modules.py:1216-1249, utils/tools.py:577-595
  a9  dur_to_mel2ph                        utils/tools.py:598-628
  a10 pitch (cwt) embedding                model/modules.py:890-948,1313-1356, utils/pitch_tools.py:27-82,258-294
  a11 energy embedding                     model/modules.py:950-960
  a12 VarianceAdaptor.forward              model/modules.py:962-1114
  a13 mel_linear                           model/CompTransTTS.py:133
  a14 PostNet                              model/modules.py:140-148
  a16 AlignmentEncoder / MAS               model/modules.py:36-75,863-888,1176-1213
  a17 liu2021 prosody encoders/predictors  model/modules.py:332-648, model/coordconv.py:33-70,140-159

Parity pinning: the reference has no tests / golden vectors, so this
restatement is pinned against the reference itself imported in the build container
(oracle/ref_import.py -> tests/golden/make_goldens.py -> tests/golden/*.npz).
The layout is [B, T, C] end to end (the reference flips between T,B,C / B,C,T).
"""
import math

import torch
import torch.nn.functional as F

F0_BIN = 256
F0_MEL_MIN = 1127 * math.log(1 + 50.0 / 700)
F0_MEL_MAX = 1127 * math.log(1 + 1100.0 / 700)


# ----------------------------------------------------------------------------- helpers
def mask_from_lengths(lengths, max_len=None):
    """True = padding.  utils/tools.py:188-196"""
    if max_len is None:
        max_len = int(lengths.max())
    ids = torch.arange(max_len, device=lengths.device)[None, :]
    return ids >= lengths[:, None]


def sinusoid_table(n_pos, dim):
    """fs2 table: [sin | cos] halves, row 0 (padding) zero.  blocks.py:66-83"""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    ang = torch.arange(n_pos, dtype=torch.float)[:, None] * freq[None, :]
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    tab[0] = 0
    return tab


def positions_from_nonpad(nonpad):
    """make_positions with padding_idx=0.  utils/tools.py:640-652"""
    m = nonpad.int()
    return (torch.cumsum(m, dim=1) * m).long()


def pos_embed(x_first_channel_or_tokens, dim):
    nonpad = x_first_channel_or_tokens.ne(0)
    pos = positions_from_nonpad(nonpad)
    tab = sinusoid_table(pos.shape[1] + 1, dim)
    return tab[pos]


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def conv1d_btc(x, w, b, pad):
    """x [B,T,Cin], w [Cout,Cin,K] (reference layout) -> [B,T,Cout]"""
    return F.conv1d(x.transpose(1, 2), w, b, padding=pad).transpose(1, 2)


# variance_predictor.ffn_act / ffn_padding of the configuration being restated (set by comp_trans_tts_forward): the FFN of every fs2
# block (transformer_fs2.py:203-239: act gelu / relu / swish, padding SAME = Conv1d(padding=k//2) / LEFT = ConstantPad1d((k-1, 0)))
# and the ConstantPad1d of the duration / pitch / energy predictors (modules.py:1270-1283,1328-1331) follow them
_SW = {"ffn_act": "gelu", "ffn_padding": "SAME"}


def conv1d_mode(x, w, b):
    """conv1d_btc with the configured ffn_padding: SAME (k-1)//2 zeros on both sides, LEFT k-1 zeros in front"""
    k = w.shape[2]
    if _SW["ffn_padding"] == "LEFT":
        return F.conv1d(F.pad(x.transpose(1, 2), (k - 1, 0)), w, b).transpose(1, 2)
    return conv1d_btc(x, w, b, (k - 1) // 2)


def ffn_act(z):
    a = _SW["ffn_act"]
    return F.gelu(z) if a == "gelu" else (torch.relu(z) if a == "relu" else (z * torch.sigmoid(z) if a == "swish" else z))


def _drop(x, p, on):
    return F.dropout(x, p, True) if (on and p > 0) else x


# ----------------------------------------------------------------------------- a8 / a9
def length_regulate_indices(dur, max_len=None):
    """Integer core of LengthRegulator (modules.py:1222-1245) + pad/crop (tools.py:577-595).

    dur [B,Ts] (any numeric dtype; truncated toward zero like int(), clamped at 0).
    Returns (idx [B,Tm] int64 = source phoneme index or -1 for zero rows,
             mel_len [B] int64 = UN-cropped sum of durations)."""
    d = dur.to(torch.float64).trunc().clamp(min=0).long()
    cum = torch.cumsum(d, dim=1)
    mel_len = cum[:, -1].clone()
    Tm = int(max_len) if max_len else int(mel_len.max())
    t = torch.arange(Tm, device=dur.device)[None, :].expand(dur.shape[0], -1).contiguous()
    idx = torch.searchsorted(cum, t, right=True)
    idx = torch.where(t < mel_len[:, None], idx, torch.full_like(idx, -1))
    return idx, mel_len


def length_regulate(x, dur, max_len=None):
    idx, mel_len = length_regulate_indices(dur, max_len)
    g = torch.gather(x, 1, idx.clamp(min=0)[..., None].expand(-1, -1, x.shape[-1]))
    out = g * (idx >= 0)[..., None].to(x.dtype)
    return out, mel_len


def dur_to_mel2ph(dur, dur_padding=None):
    """utils/tools.py:598-628 - 1-based phoneme index per frame, 0 for pad; width = max sum."""
    d = torch.round(dur.float()).long()
    if dur_padding is not None:
        d = d * (1 - dur_padding.long())
    cum = torch.cumsum(d, 1)
    total = cum[:, -1]
    Tm = int(total.max())
    t = torch.arange(Tm, device=dur.device)[None, :].expand(dur.shape[0], -1).contiguous()
    idx = torch.searchsorted(cum, t, right=True) + 1
    return torch.where(t < total[:, None], idx, torch.zeros_like(idx))


# ----------------------------------------------------------------------------- pitch helpers
def cwt2f0_norm(cwt_spec, mean, std, width, eps=1e-9):
    """utils/pitch_tools.py:258-294 (pitch_norm == 'log')."""
    b = (torch.arange(cwt_spec.shape[-1], dtype=torch.float, device=cwt_spec.device) + 3.5) ** (-2.5)
    rec = (cwt_spec * b[None, None, :]).sum(-1)
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
    f0_mel = torch.where(f0_mel <= 1, torch.ones_like(f0_mel), f0_mel)
    f0_mel = torch.where(f0_mel > F0_BIN - 1, torch.full_like(f0_mel, F0_BIN - 1), f0_mel)
    return (f0_mel + 0.5).long()


# ----------------------------------------------------------------------------- a4-a6
def fft_blocks(sd, pre, x, pad_mask, n_layers, n_heads, ksize, p_drop, use_pos, train_dropout=False, taps=None):
    B, T, C = x.shape
    nonpad = (~pad_mask).to(x.dtype)[..., None]
    if use_pos:
        x = x + sd[pre + "pos_embed_alpha"] * pos_embed(x[..., 0], C)
        x = _drop(x, p_drop, train_dropout)
    x = x * nonpad
    dh = C // n_heads
    for l in range(n_layers):
        p = f"{pre}layers.{l}.op."
        res = x
        h = layer_norm(x, sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], 1e-12)
        qkv = h @ sd[p + "self_attn.in_proj_weight"].t()
        q, k, v = qkv.split(C, dim=-1)
        q = q.reshape(B, T, n_heads, dh).transpose(1, 2) * dh ** -0.5
        k = k.reshape(B, T, n_heads, dh).transpose(1, 2)
        v = v.reshape(B, T, n_heads, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        s = s.masked_fill(pad_mask[:, None, None, :], float("-inf"))
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, C) @ sd[p + "self_attn.out_proj.weight"].t()
        x = (res + _drop(a, p_drop, train_dropout)) * nonpad
        res = x
        h = layer_norm(x, sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], 1e-12)
        f1 = "ffn.ffn_1.1." if _SW["ffn_padding"] == "LEFT" else "ffn.ffn_1."          # LEFT: nn.Sequential(ConstantPad1d, Conv1d)
        z = conv1d_mode(h, sd[p + f1 + "weight"], sd[p + f1 + "bias"]) * ksize ** -0.5
        g = _drop(ffn_act(z), p_drop, train_dropout)
        y = g @ sd[p + "ffn.ffn_2.weight"].t() + sd[p + "ffn.ffn_2.bias"]
        x = (res + _drop(y, p_drop, train_dropout)) * nonpad
        if taps is not None:
            taps[f"{pre}layer{l}"] = x
    x = layer_norm(x, sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5) * nonpad
    return x


def text_encoder(sd, cfg, tokens, pad_mask, train_dropout=False, taps=None):
    c = cfg["transformer_fs2"]
    H = c["encoder_hidden"]
    emb = math.sqrt(H) * F.embedding(tokens, sd["encoder.embed_tokens.weight"], padding_idx=0)
    x = emb + pos_embed(tokens, H)
    x = _drop(x, c["encoder_dropout"], train_dropout)
    x = fft_blocks(sd, "encoder.", x, pad_mask, c["encoder_layer"], c["encoder_head"], c["ffn_kernel_size"],
                   c["encoder_dropout"], False, train_dropout, taps)
    return x, emb


# ----------------------------------------------------------------------------- a7, a10, a11 predictors
def _predictor_convs(sd, pre, x, n_layers, ksize, p_drop, mask_nonpad, train_dropout):
    for i in range(n_layers):
        x = conv1d_mode(x, sd[f"{pre}conv.{i}.1.weight"], sd[f"{pre}conv.{i}.1.bias"])
        x = torch.relu(x)
        x = layer_norm(x, sd[f"{pre}conv.{i}.3.weight"], sd[f"{pre}conv.{i}.3.bias"], 1e-12)
        x = _drop(x, p_drop, train_dropout)
        if mask_nonpad is not None:
            x = x * mask_nonpad
    return x


def duration_predictor(sd, cfg, x, src_pad, train_dropout=False):
    vp = cfg["variance_predictor"]
    nonpad = (~src_pad).to(x.dtype)[..., None]
    pre = "variance_adaptor.duration_predictor."
    h = _predictor_convs(sd, pre, x, vp["dur_predictor_layers"], vp["dur_predictor_kernel"], vp["dropout"], nonpad,
                         train_dropout)
    out = h @ sd[pre + "linear.weight"].t() + sd[pre + "linear.bias"]
    return (out * nonpad).squeeze(-1)


def pitch_like_predictor(sd, pre, cfg, x, train_dropout=False):
    vp = cfg["variance_predictor"]
    x = x + sd[pre + "pos_embed_alpha"] * pos_embed(x[..., 0], x.shape[-1])
    h = _predictor_convs(sd, pre, x, vp["predictor_layers"], vp["predictor_kernel"], vp["dropout"], None, train_dropout)
    return h @ sd[pre + "linear.weight"].t() + sd[pre + "linear.bias"]


def _grad_scale(x, g):
    return x.detach() + g * (x - x.detach())


# ----------------------------------------------------------------------------- a12
def mas_width1(attn_map):
    """model/modules.py:36-64 (numba-jitted in the reference; plain numpy here).  attn_map [T1 (mel), T2 (text)] float32."""
    import numpy as np
    opt = np.zeros_like(attn_map)
    with np.errstate(divide="ignore"):
        attn_map = np.log(attn_map)
    attn_map[0, 1:] = -np.inf
    log_p = np.zeros_like(attn_map)
    log_p[0, :] = attn_map[0, :]
    prev_ind = np.zeros_like(attn_map, dtype=np.int64)
    for i in range(1, attn_map.shape[0]):
        prev = log_p[i - 1]
        left = np.concatenate([[-np.inf], prev[:-1]]).astype(prev.dtype)
        take_left = left >= prev
        take_left[0] = False
        log_p[i] = attn_map[i] + np.where(take_left, left, prev)
        prev_ind[i] = np.arange(attn_map.shape[1]) - take_left.astype(np.int64)
    cur = attn_map.shape[1] - 1
    for i in range(attn_map.shape[0] - 1, -1, -1):
        opt[i, cur] = 1
        cur = prev_ind[i, cur]
    opt[0, cur] = 1
    return opt


def binarize_attention(attn_soft, in_lens, out_lens):
    """model/modules.py:66-75,863-872 b_mas over the batch.  attn_soft [B,1,Tm,Ts] -> hard 0/1 of the same shape."""
    import numpy as np
    a = attn_soft.detach().cpu().numpy()
    out = np.zeros_like(a)
    for b in range(a.shape[0]):
        o, i = int(out_lens[b]), int(in_lens[b])
        out[b, 0, :o, :i] = mas_width1(a[b, 0, :o, :i].copy())
    return torch.from_numpy(out)


def alignment_encoder(sd, mel, text_emb, src_pad, attn_prior, temperature, speaker_embedding=None):
    """model/modules.py:1176-1213 AlignmentEncoder.forward.  mel [B,Tm,80], text_emb [B,Ts,256], attn_prior [B,Tm,Ts]
    -> (attn_soft [B,1,Tm,Ts], attn_logprob [B,1,Tm,Ts]).  multi_speaker: bias-free projections of the speaker embedding are added
    to every key / query position (:1188-1194)."""
    a = "variance_adaptor.aligner."
    if speaker_embedding is not None:
        text_emb = text_emb + (speaker_embedding @ sd[a + "key_spk_proj.linear.weight"].t())[:, None, :]
        mel = mel + (speaker_embedding @ sd[a + "query_spk_proj.linear.weight"].t())[:, None, :]
    k = torch.relu(conv1d_btc(text_emb, sd[a + "key_proj.0.conv.weight"], sd[a + "key_proj.0.conv.bias"], 1))
    k = conv1d_btc(k, sd[a + "key_proj.2.conv.weight"], sd[a + "key_proj.2.conv.bias"], 0)
    q = torch.relu(conv1d_btc(mel, sd[a + "query_proj.0.conv.weight"], sd[a + "query_proj.0.conv.bias"], 1))
    q = torch.relu(conv1d_btc(q, sd[a + "query_proj.2.conv.weight"], sd[a + "query_proj.2.conv.bias"], 0))
    q = conv1d_btc(q, sd[a + "query_proj.4.conv.weight"], sd[a + "query_proj.4.conv.bias"], 0)
    attn = -temperature * ((q[:, :, None, :] - k[:, None, :, :]) ** 2).sum(-1)          # [B,Tm,Ts]
    attn = torch.log_softmax(attn, dim=-1) + torch.log(attn_prior + 1e-8)
    logprob = attn.clone()
    attn = attn.masked_fill(src_pad[:, None, :], float("-inf"))
    return torch.softmax(attn, dim=-1)[:, None], logprob[:, None]


def phoneme_level_pitch(n_phones, src_lens, mel2ph, mel_lens, pitch_frame):
    """utils/tools.py:47-53 + model/modules.py:873-880: per utterance, scatter_add the first mel_len frames of the f0 contour at
    mel2ph - 1 into src_len slots and divide by the (clamped) frame counts; right-padded to the longest utterance."""
    B = pitch_frame.shape[0]
    out = torch.zeros(B, int(src_lens.max()))
    for b in range(B):
        s, m = int(src_lens[b]), int(mel_lens[b])
        idx = mel2ph[b, :m].long() - 1
        tot = torch.zeros(s).scatter_add(0, idx, pitch_frame[b, :m].float())
        num = torch.zeros(s).scatter_add(0, idx, torch.ones(m)).clamp_min(1)
        out[b, :s] = tot / num
    return out


def denorm_f0(f0, uv, pitch_cfg, pitch_padding=None):
    """utils/pitch_tools.py:69-82"""
    if pitch_cfg["pitch_norm"] == "standard":
        f0 = f0 * pitch_cfg["f0_std"] + pitch_cfg["f0_mean"]
    if pitch_cfg["pitch_norm"] == "log":
        f0 = 2 ** f0
    if uv is not None and pitch_cfg["use_uv"]:
        f0[uv > 0] = 0
    if pitch_padding is not None:
        f0[pitch_padding] = 0
    return f0


def phoneme_level_energy(dur, src_lens, energy_frame):
    """utils/tools.py:56-66 + model/modules.py:882-888: per-phoneme mean of the frame-level energy."""
    import numpy as np
    B, Ts = dur.shape
    out = torch.zeros(B, int(src_lens.max()))
    d_np, e_np = dur.int().cpu().numpy(), energy_frame.cpu().numpy()
    for b in range(B):
        e = e_np[b].copy()
        pos = 0
        for i in range(int(src_lens[b])):
            d = int(d_np[b, i])
            e[i] = e[pos:pos + d].mean() if d > 0 else 0
            pos += d
        out[b, : int(src_lens[b])] = torch.from_numpy(e[: int(src_lens[b])])
    return out


# ============================================================================= a17: liu2021 implicit prosody modelling
def coord_channels(T, W):
    """AddCoords rank 2, with_r (coordconv.py:33-70): planes xx (varies along dim_y = time), yy (varies along dim_x = mel bin)
    in [-1,1] and rr = sqrt((xx-.5)^2 + (yy-.5)^2), each [T,W] float32.  NOTE: T is the PADDED batch length."""
    xx = torch.arange(T, dtype=torch.int32)[:, None].expand(T, W).float() / (T - 1)
    yy = torch.arange(W, dtype=torch.int32)[None, :].expand(T, W).float() / (W - 1)
    xx = xx * 2 - 1
    yy = yy * 2 - 1
    rr = torch.sqrt(torch.pow(xx - 0.5, 2) + torch.pow(yy - 0.5, 2))
    return xx, yy, rr


def gru_sequence(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """single-layer nn.GRU, batch_first, zero initial state (gate order r|z|n; n = tanh(gi_n + r*(W_hn h + b_hn))).
    Runs over ALL T steps, pads included - the reference never packs (modules.py:390-391, 636-637)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1), h


def _bn_nd(sd, p, x, cdim, training_bn, new_stats):
    """BatchNorm over every dim but `cdim` (batch statistics incl. pads when training_bn)."""
    C = x.shape[cdim]
    shape = [1] * x.dim()
    shape[cdim] = C
    if training_bn:
        flat = x.transpose(cdim, -1).reshape(-1, C)
        mean, var = flat.mean(0), flat.var(0, unbiased=False)
        if new_stats is not None:
            nrow = flat.shape[0]
            new_stats[p + "running_mean"] = 0.9 * sd[p + "running_mean"] + 0.1 * mean.detach()
            new_stats[p + "running_var"] = 0.9 * sd[p + "running_var"] + 0.1 * var.detach() * nrow / (nrow - 1)
    else:
        mean, var = sd[p + "running_mean"], sd[p + "running_var"]
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + 1e-5) * sd[p + "weight"].view(shape) + sd[p + "bias"].view(shape)


def reference_encoder(sd, p, mel, mel_pad, training_bn, new_stats=None):
    """ReferenceEncoder.forward (modules.py:371-392): 6x [Conv2d 3x3 stride (1,2) pad 1 -> BN2d -> ReLU], the first conv on
    [mel | xx | yy | rr]; -> [N,T,C*W'] (channel-major) -> pad rows zeroed -> GRU.  Returns (memory [N,T,G], last [N,G])."""
    N, T, W = mel.shape
    xx, yy, rr = coord_channels(T, W)
    out = torch.stack([mel, xx.expand(N, T, W), yy.expand(N, T, W), rr.expand(N, T, W)], 1)      # [N,4,T,W]
    i = 0
    while (p + f"bns.{i}.weight") in sd:
        wkey = p + (f"convs.{i}.conv." if i == 0 else f"convs.{i}.")
        out = F.conv2d(out, sd[wkey + "weight"], sd[wkey + "bias"], stride=(1, 2), padding=(1, 1))
        out = torch.relu(_bn_nd(sd, p + f"bns.{i}.", out, 1, training_bn, new_stats))
        i += 1
    out = out.transpose(1, 2).contiguous().view(N, T, -1)
    if mel_pad is not None:
        out = out.masked_fill(mel_pad.unsqueeze(-1), 0)
    g = p + "gru."
    return gru_sequence(out, sd[g + "weight_ih_l0"], sd[g + "weight_hh_l0"], sd[g + "bias_ih_l0"], sd[g + "bias_hh_l0"])


def utterance_prosody_encoder(sd, p, mel, mel_pad, E, training_bn, new_stats=None):
    """UtteranceLevelProsodyEncoder.forward (modules.py:555-569) + STL / StyleEmbedAttention with one head (:453-534)."""
    _, last = reference_encoder(sd, p + "encoder.", mel, mel_pad, training_bn, new_stats)
    ep = last @ sd[p + "encoder_prj.weight"].t() + sd[p + "encoder_prj.bias"]                 # [N,E/2]
    keys_in = torch.tanh(sd[p + "stl.embed"])                                                   # [tokens,E]
    a = p + "stl.attention."
    q = ep @ sd[a + "W_query.weight"].t()
    k = keys_in @ sd[a + "W_key.weight"].t()
    v = keys_in @ sd[a + "W_value.weight"].t()
    sc = torch.softmax(q @ k.t() / (E ** 0.5), -1)                                              # key_dim = E // num_heads = E
    out = (sc @ v) @ sd[p + "encoder_bottleneck.weight"].t() + sd[p + "encoder_bottleneck.bias"]
    return out.unsqueeze(1)                                                                     # [N,1,bottleneck_u]; dropout p = 0.


def phoneme_prosody_encoder(sd, p, x, src_pad, mel, mel_pad, E, training_bn, new_stats=None):
    """PhonemeLevelProsodyEncoder.forward (modules.py:420-450)."""
    mem, _ = reference_encoder(sd, p + "encoder.", mel, mel_pad, training_bn, new_stats)
    ep = mem @ sd[p + "encoder_prj.weight"].t() + sd[p + "encoder_prj.bias"]                  # [N,Tm,2E]
    k, v = ep[..., :E], ep[..., E:]
    q = x @ sd[p + "linears.0.linear.weight"].t()
    k = k @ sd[p + "linears.1.linear.weight"].t()
    attn = q @ k.transpose(1, 2) / math.sqrt(E)
    attn = attn.masked_fill(mel_pad.unsqueeze(1), float("-inf"))
    attn = torch.softmax(attn, -1).masked_fill(src_pad.unsqueeze(-1), 0.0)
    out = torch.bmm(attn, v) @ sd[p + "encoder_bottleneck.weight"].t() + sd[p + "encoder_bottleneck.bias"]
    return out.masked_fill(src_pad.unsqueeze(-1), 0.0), attn


def parallel_prosody_predictor(sd, p, x, E, phoneme_level, p_drop, train_dropout=False):
    """ParallelProsodyPredictor.forward (modules.py:626-648): 2x [Conv1d k3 -> ReLU -> LayerNorm(1e-5) -> dropout] -> bi-GRU."""
    for i in (1, 2):
        w, b = sd[p + f"conv_layer.conv1d_{i}.conv.weight"], sd[p + f"conv_layer.conv1d_{i}.conv.bias"]
        x = torch.relu(conv1d_btc(x, w, b, 1 if i == 2 else (w.shape[2] - 1) // 2))
        x = layer_norm(x, sd[p + f"conv_layer.layer_norm_{i}.weight"], sd[p + f"conv_layer.layer_norm_{i}.bias"], 1e-5)
        x = _drop(x, p_drop, train_dropout)
    g = p + "gru."
    mf, hf = gru_sequence(x, sd[g + "weight_ih_l0"], sd[g + "weight_hh_l0"], sd[g + "bias_ih_l0"], sd[g + "bias_hh_l0"])
    mb, hb = gru_sequence(x, sd[g + "weight_ih_l0_reverse"], sd[g + "weight_hh_l0_reverse"], sd[g + "bias_ih_l0_reverse"],
                          sd[g + "bias_hh_l0_reverse"], reverse=True)
    pv = torch.cat([mf, mb], -1) if phoneme_level else torch.cat([hf, hb], -1).unsqueeze(1)
    return pv @ sd[p + "predictor_bottleneck.weight"].t() + sd[p + "predictor_bottleneck.bias"]


def liu2021_prosody(sd, cfg, x, src_pad, mel, mel_pad, training, train_dropout=False, new_stats=None):
    """VarianceAdaptor liu2021 branch (modules.py:1002-1022).  Returns (x, prosody_info)."""
    va = "variance_adaptor."
    E = cfg["transformer"]["encoder_hidden"]
    pd = cfg["prosody_modeling"]["liu2021"]["predictor_dropout"]
    up_emb = pp_emb = pp_attn = None
    if training:
        up_emb = utterance_prosody_encoder(sd, va + "utterance_prosody_encoder.", mel, mel_pad, E, True, new_stats)
        pp_emb, pp_attn = phoneme_prosody_encoder(sd, va + "phoneme_prosody_encoder.", x, src_pad, mel, mel_pad, E, True, new_stats)
    up_vec = parallel_prosody_predictor(sd, va + "utterance_prosody_predictor.", x, E, False, pd, train_dropout)
    u = up_emb if training else up_vec
    x = x + (u @ sd[va + "utterance_prosody_prj.weight"].t() + sd[va + "utterance_prosody_prj.bias"])
    pp_vec = parallel_prosody_predictor(sd, va + "phoneme_prosody_predictor.", x, E, True, pd, train_dropout)
    pp = pp_emb if training else pp_vec
    x = x + (pp @ sd[va + "phoneme_prosody_prj.weight"].t() + sd[va + "phoneme_prosody_prj.bias"])
    return x, (up_emb, pp_emb, up_vec, pp_vec, pp_attn)


def variance_adaptor(sd, cfg, pre_cfg, text, src_lens, src_pad, mel_lens, mel_pad, max_mel_len,
                     p_targets, e_targets, d_targets, speaker_embedding=None,
                     p_control=1.0, e_control=1.0, d_control=1.0, train_dropout=False, taps=None,
                     text_embedding=None, mel=None, attn_prior=None, step=None, bin_start_steps=None, attn_out=None,
                     training=False, new_stats=None, prosody_out=None):
    vp = cfg["variance_predictor"]
    va = "variance_adaptor."
    pitch_cfg = pre_cfg["preprocessing"]["pitch"]
    pitch_type = pitch_cfg["pitch_type"]
    assert pitch_type in ("cwt", "frame", "ph") and pitch_cfg["pitch_norm"] in ("log", "standard") and not pitch_cfg.get("pitch_ar", False)
    assert pitch_type != "cwt" or (pitch_cfg["pitch_norm"] == "log" and pitch_cfg["use_uv"])
    frame_energy = pre_cfg["preprocessing"]["energy"]["feature"] == "frame_level"
    x = text.clone()
    if speaker_embedding is not None:
        x = x + speaker_embedding[:, None, :]
    if cfg["prosody_modeling"]["model_type"] == "liu2021":
        x, pros = liu2021_prosody(sd, cfg, x, src_pad, mel, mel_pad, training, train_dropout, new_stats)
        if prosody_out is not None:
            prosody_out.append(pros)
    log_d = duration_predictor(sd, cfg, _grad_scale(x, vp["predictor_grad"]), src_pad, train_dropout)
    x_org = x
    mel2ph_inf = None
    if attn_prior is not None:         # unsupervised duration modelling (modules.py:1031-1053)
        attn_soft, attn_logprob = alignment_encoder(sd, mel, text_embedding, src_pad, attn_prior.transpose(1, 2),
                                                    cfg["duration_modeling"]["aligner_temperature"], speaker_embedding)
        attn_hard = binarize_attention(attn_soft, src_lens, mel_lens)
        attn_hard_dur = attn_hard.sum(2)[:, 0, :]
        if attn_out is not None:
            attn_out.extend([attn_soft, attn_hard, attn_hard_dur, attn_logprob])
        if step < bin_start_steps:
            x = torch.bmm(attn_soft.squeeze(1), x)
            mel_len = mel_lens
        else:
            x, mel_len = length_regulate(x, attn_hard_dur, max_mel_len)
        d_rounded = attn_hard_dur
        p_targets = dict(p_targets)
        p_targets["mel2ph"] = dur_to_mel2ph(d_rounded, src_pad)[:, :max_mel_len]
        if not frame_energy:
            e_targets = phoneme_level_energy(attn_hard_dur, src_lens, e_targets)
    elif d_targets is not None:
        x, mel_len = length_regulate(x, d_targets, max_mel_len)
        d_rounded = d_targets
    else:
        d_rounded = torch.clamp(torch.round(torch.exp(log_d) - 1) * d_control, min=0)
        x, mel_len = length_regulate(x, d_rounded, max_mel_len)
        mel_pad = mask_from_lengths(mel_len)
        mel2ph_inf = dur_to_mel2ph(d_rounded, src_pad)
    if taps is not None:
        taps["lr_out"] = x

    ve = cfg.get("variance_embedding", {})
    use_pitch, use_energy = ve.get("use_pitch_embed", True), ve.get("use_energy_embed", True)      # modules.py:735-736,1071,1092-1095
    out = x
    p_pred = e_pred = None
    if use_pitch and pitch_type == "frame":
        # pitch (frame)   modules.py:906-938: [f0, uv logit] per frame from the regulated sequence
        pitch_pred = pitch_like_predictor(sd, va + "pitch_predictor.", cfg, _grad_scale(x, vp["predictor_grad"]), train_dropout) * p_control
        if p_targets is not None:
            p_targets = dict(p_targets)
            mel2ph = p_targets["mel2ph"]
            f0, uv = p_targets["f0"].clone(), p_targets["uv"]
        else:
            mel2ph = mel2ph_inf
            f0 = pitch_pred[:, :, 0]
            uv = (pitch_pred[:, :, 1] > 0) if pitch_cfg["use_uv"] else None
        pad = mel2ph[:, : f0.shape[1]] == 0
        f0_denorm = denorm_f0(f0, uv, pitch_cfg, pad)
        f0[pad] = 0                                            # in place: the target / (inference) channel 0 of the prediction
        if p_targets is not None:
            p_targets["f0"] = f0
        pitch_emb = F.embedding(f0_to_coarse(f0_denorm), sd[va + "pitch_embed.weight"], padding_idx=0)
        p_pred = {"pitch_pred": pitch_pred, "f0_denorm": f0_denorm, "cwt": None, "f0_mean": None, "f0_std": None}
        out = out + pitch_emb
    elif use_pitch and pitch_type == "ph":
        # pitch (ph)   modules.py:892-905,1083-1084: one f0 per phoneme from the encoder side, gathered to frames through mel2ph
        pitch_pred = pitch_like_predictor(sd, va + "pitch_predictor.", cfg, _grad_scale(x_org, vp["predictor_grad"]), train_dropout) * p_control
        if p_targets is not None:
            p_targets = dict(p_targets)
            mel2ph = p_targets["mel2ph"]
            p_targets["f0"] = phoneme_level_pitch(x_org.shape[1], src_lens, mel2ph, mel_len, p_targets["f0"])
            f0 = p_targets["f0"]
        else:
            mel2ph = mel2ph_inf
            f0 = pitch_pred[:, :, 0]
        f0_denorm = denorm_f0(f0, None, pitch_cfg, x_org.sum().abs() == 0)
        ph_ids = F.pad(f0_to_coarse(f0_denorm), [1, 0])
        pitch_emb = F.embedding(torch.gather(ph_ids, 1, mel2ph.long()), sd[va + "pitch_embed.weight"], padding_idx=0)
        p_pred = {"pitch_pred": pitch_pred, "f0_denorm": f0_denorm, "cwt": None, "f0_mean": None, "f0_std": None}
        out = out + pitch_emb
    elif use_pitch:
        # pitch (cwt)
        dec_inp = _grad_scale(x, vp["predictor_grad"])
        h = dec_inp @ sd[va + "cwt_predictor.0.weight"].t() + sd[va + "cwt_predictor.0.bias"]
        cwt = pitch_like_predictor(sd, va + "cwt_predictor.1.", cfg, h, train_dropout) * p_control
        s = x_org[:, 0, :]
        s = torch.relu(s @ sd[va + "cwt_stats_layers.0.weight"].t() + sd[va + "cwt_stats_layers.0.bias"])
        s = torch.relu(s @ sd[va + "cwt_stats_layers.2.weight"].t() + sd[va + "cwt_stats_layers.2.bias"])
        stats = s @ sd[va + "cwt_stats_layers.4.weight"].t() + sd[va + "cwt_stats_layers.4.bias"]
        f0_mean, f0_std = stats[:, 0], stats[:, 1]
        if p_targets is not None:
            p_targets = dict(p_targets)
            mel2ph = p_targets["mel2ph"]
            p_targets["f0"] = cwt2f0_norm(p_targets["cwt_spec"], p_targets["f0_mean"], p_targets["f0_std"],
                                          mel2ph.shape[1], pitch_cfg["pitch_norm_eps"])
            p_targets["f0_cwt"] = p_targets["f0"]
            f0, uv = p_targets["f0"], p_targets["uv"]
        else:
            f0 = cwt2f0_norm(cwt[:, :, :10], f0_mean, f0_std * vp["cwt_std_scale"], mel2ph_inf.shape[1],
                             pitch_cfg["pitch_norm_eps"])
            uv = cwt[:, :, -1] > 0
        f0_denorm = 2 ** f0
        f0_denorm = torch.where(uv > 0, torch.zeros_like(f0_denorm), f0_denorm)
        pitch_emb = F.embedding(f0_to_coarse(f0_denorm), sd[va + "pitch_embed.weight"], padding_idx=0)
        p_pred = {"pitch_pred": None, "f0_denorm": f0_denorm, "cwt": cwt, "f0_mean": f0_mean, "f0_std": f0_std}
        out = out + pitch_emb
    if use_energy:
        # energy (phoneme level; NOTE modules.py:951 discards the grad-scaled tensor -> full gradient)
        e_pred = pitch_like_predictor(sd, va + "energy_predictor.", cfg, x if frame_energy else x_org, train_dropout).squeeze(-1)
        bins = sd[va + "energy_bins"]
        if e_targets is not None:
            e_idx = torch.bucketize(e_targets, bins)
        else:
            e_pred = e_pred * e_control
            e_idx = torch.bucketize(e_pred, bins)
        e_emb = F.embedding(e_idx, sd[va + "energy_embedding.weight"], padding_idx=0)
        if frame_energy:                     # modules.py:1092-1094
            out = out + e_emb
        else:
            e_emb_frames, _ = length_regulate(e_emb, d_rounded, max_mel_len)
            out = out + e_emb_frames
    if taps is not None:
        taps["va_out"] = out
    return out, p_targets, p_pred, e_targets, e_pred, log_d, d_rounded, mel_len, mel_pad


# ----------------------------------------------------------------------------- a14
def postnet(sd, x, training_bn, train_dropout=False, new_stats=None):
    """x [B,T,80] -> [B,T,80]; BN over all B*T positions incl. pads (modules.py:140-148)."""
    n = 5
    for i in range(n):
        p = f"postnet.convolutions.{i}."
        x = conv1d_btc(x, sd[p + "0.conv.weight"], sd[p + "0.conv.bias"], 2)
        C = x.shape[-1]
        if training_bn:
            flat = x.reshape(-1, C)
            mean = flat.mean(0)
            var = flat.var(0, unbiased=False)
            if new_stats is not None:
                m = 0.1
                nrow = flat.shape[0]
                new_stats[p + "1.running_mean"] = (1 - m) * sd[p + "1.running_mean"] + m * mean.detach()
                new_stats[p + "1.running_var"] = (1 - m) * sd[p + "1.running_var"] + m * var.detach() * nrow / (nrow - 1)
        else:
            mean, var = sd[p + "1.running_mean"], sd[p + "1.running_var"]
        x = (x - mean) / torch.sqrt(var + 1e-5) * sd[p + "1.weight"] + sd[p + "1.bias"]
        if i < n - 1:
            x = torch.tanh(x)
        x = _drop(x, 0.5, train_dropout)
    return x


# ----------------------------------------------------------------------------- a1
def comp_trans_tts_forward(sd, model_cfg, pre_cfg, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None,
                           max_mel_len=None, p_targets=None, e_targets=None, d_targets=None, attn_priors=None,
                           spker_embeds=None, p_control=1.0, e_control=1.0, d_control=1.0, step=None,
                           training=False, train_dropout=False, taps=None, new_stats=None, bin_start_steps=6000):
    """Restates model/CompTransTTS.py:64-152 for block_type == transformer_fs2,
    learn_alignment == False.  `training` selects BatchNorm batch statistics;
    `train_dropout` additionally turns the dropouts on (for CPU-baseline timing)."""
    assert model_cfg["block_type"] == "transformer_fs2"
    vp_sw = model_cfg.get("variance_predictor", {})
    _SW["ffn_act"], _SW["ffn_padding"] = vp_sw.get("ffn_act", "gelu"), vp_sw.get("ffn_padding", "SAME")
    src_pad = mask_from_lengths(src_lens, max_src_len)
    mel_pad = mask_from_lengths(mel_lens, max_mel_len) if mel_lens is not None else None
    enc, text_emb = text_encoder(sd, model_cfg, texts, src_pad, train_dropout, taps)
    if taps is not None:
        taps["encoder_out"] = enc
    spk = None
    if model_cfg["multi_speaker"]:
        if "speaker_emb.bias" in sd:
            spk = spker_embeds @ sd["speaker_emb.weight"].t() + sd["speaker_emb.bias"]
        else:
            spk = F.embedding(speakers, sd["speaker_emb.weight"])
    attn_out, prosody_out = [], []
    (x, p_targets, p_pred, e_targets, e_pred, log_d, d_rounded, mel_lens, mel_pad) = variance_adaptor(
        sd, model_cfg, pre_cfg, enc, src_lens, src_pad, mel_lens, mel_pad, max_mel_len, p_targets, e_targets,
        d_targets, spk, p_control, e_control, d_control, train_dropout, taps, text_embedding=text_emb, mel=mels,
        attn_prior=attn_priors, step=step, bin_start_steps=bin_start_steps, attn_out=attn_out, training=training,
        new_stats=new_stats, prosody_out=prosody_out)
    c = model_cfg["transformer_fs2"]
    dec = fft_blocks(sd, "decoder.", x, mel_pad, c["decoder_layer"], c["decoder_head"], c["ffn_kernel_size"],
                     c["decoder_dropout"], True, train_dropout, taps)
    if taps is not None:
        taps["decoder_out"] = dec
    mel = dec @ sd["mel_linear.weight"].t() + sd["mel_linear.bias"]
    post = postnet(sd, mel, training, train_dropout, new_stats) + mel
    return (mel, post, p_pred, e_pred, log_d, d_rounded, src_pad, mel_pad, src_lens, mel_lens,
            tuple(attn_out) if attn_out else (None, None, None, None), prosody_out[0] if prosody_out else None, p_targets, e_targets)


# ============================================================================= a15: conformer plugin
def interleaved_sinusoid_table(n_position, d_hid):
    """blocks.py:26-46 get_sinusoid_encoding_table: angle = pos / 10000^(2*(j//2)/d), sin on even j, cos on odd j
    (computed in float64 numpy, stored as float32)."""
    import numpy as np
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab = np.array(ang)
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).float()


def _swish(x):
    return x * torch.sigmoid(x)


def _relative_shift(ps):
    """conformer.py:423-431 (Transformer-XL shift)"""
    B, H, T1, T2 = ps.shape
    padded = torch.cat([ps.new_zeros(B, H, T1, 1), ps], dim=-1).view(B, H, T2 + 1, T1)
    return padded[:, :, 1:].reshape(B, H, T1, T2)


def conformer_stack(sd, pre, x, pad_mask, n_layers, n_heads, p_drop, training_bn, train_dropout=False, new_stats=None, taps=None):
    """conformer.py:162-246 ConformerBlock x n_layers.  NOTE the attention receives no mask (nn.Sequential drops it,
    conformer.py:243 vs :326) and scores are scaled by sqrt(d_model) (:375,:409)."""
    B, T, C = x.shape
    dh = C // n_heads
    pos_table = sd[pre + "position_enc"][0, :T]
    for l in range(n_layers):
        p = f"{pre}layer_stack.{l}.sequential."
        for ff in ("0", "3"):       # half-step feed-forward modules
            q = p + ff + ".module.sequential."
            h = layer_norm(x, sd[q + "0.weight"], sd[q + "0.bias"], 1e-5)
            h = _drop(_swish(h @ sd[q + "1.linear.weight"].t() + sd[q + "1.linear.bias"]), p_drop, train_dropout)
            h = _drop(h @ sd[q + "4.linear.weight"].t() + sd[q + "4.linear.bias"], p_drop, train_dropout)
            x = x + 0.5 * h
            if ff == "3":
                break
            # multi-headed self attention with relative positions
            a = p + "1.module."
            h = layer_norm(x, sd[a + "layer_norm.weight"], sd[a + "layer_norm.bias"], 1e-5)
            qh = (h @ sd[a + "attention.query_proj.linear.weight"].t()).view(B, T, n_heads, dh)
            kh = (h @ sd[a + "attention.key_proj.linear.weight"].t()).view(B, T, n_heads, dh).permute(0, 2, 1, 3)
            vh = (h @ sd[a + "attention.value_proj.linear.weight"].t()).view(B, T, n_heads, dh).permute(0, 2, 1, 3)
            ph = (pos_table @ sd[a + "attention.pos_proj.linear.weight"].t()).view(T, n_heads, dh)
            content = (qh + sd[a + "attention.u_bias"]).transpose(1, 2) @ kh.transpose(2, 3)
            pscore = (qh + sd[a + "attention.v_bias"]).transpose(1, 2) @ ph.permute(1, 2, 0)[None]
            score = (content + _relative_shift(pscore)) / math.sqrt(C)
            attn = _drop(torch.softmax(score, -1), p_drop, train_dropout)
            ctxv = (attn @ vh).transpose(1, 2).reshape(B, T, C)
            x = x + _drop(ctxv @ sd[a + "attention.out_proj.linear.weight"].t(), p_drop, train_dropout)
            # convolution module
            c = p + "2.module.sequential."
            h = layer_norm(x, sd[c + "0.weight"], sd[c + "0.bias"], 1e-5)
            h = h @ sd[c + "2.conv.weight"][:, :, 0].t() + sd[c + "2.conv.bias"]
            h = h[..., :C] * torch.sigmoid(h[..., C:])
            wdw = sd[c + "4.conv.weight"]
            h = F.conv1d(h.transpose(1, 2), wdw, None, padding=(wdw.shape[-1] - 1) // 2, groups=C).transpose(1, 2)
            if training_bn:
                flat = h.reshape(-1, C)
                mean, var = flat.mean(0), flat.var(0, unbiased=False)
                if new_stats is not None:
                    n = flat.shape[0]
                    new_stats[c + "5.running_mean"] = 0.9 * sd[c + "5.running_mean"] + 0.1 * mean.detach()
                    new_stats[c + "5.running_var"] = 0.9 * sd[c + "5.running_var"] + 0.1 * var.detach() * n / (n - 1)
            else:
                mean, var = sd[c + "5.running_mean"], sd[c + "5.running_var"]
            h = _swish((h - mean) / torch.sqrt(var + 1e-5) * sd[c + "5.weight"] + sd[c + "5.bias"])
            h = _drop(h @ sd[c + "7.conv.weight"][:, :, 0].t() + sd[c + "7.conv.bias"], p_drop, train_dropout)
            x = x + h
        x = layer_norm(x, sd[p + "4.weight"], sd[p + "4.bias"], 1e-5)
        x = x.masked_fill(pad_mask[..., None], 0)
        if taps is not None:
            taps[f"{pre}layer{l}"] = x
    return x


def comp_trans_tts_forward_conformer(sd, model_cfg, pre_cfg, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None,
                                     max_mel_len=None, p_targets=None, e_targets=None, d_targets=None, attn_priors=None,
                                     spker_embeds=None, p_control=1.0, e_control=1.0, d_control=1.0, step=None,
                                     training=False, train_dropout=False, taps=None, new_stats=None):
    """model/CompTransTTS.py:64-152 with block_type == conformer (conformer.py:20-159 encoder/decoder wrappers)."""
    assert model_cfg["block_type"] == "conformer" and attn_priors is None and not model_cfg["multi_speaker"]
    vp_sw = model_cfg.get("variance_predictor", {})
    _SW["ffn_act"], _SW["ffn_padding"] = vp_sw.get("ffn_act", "gelu"), vp_sw.get("ffn_padding", "SAME")       # the predictors' padding
    c = model_cfg["conformer"]
    src_pad = mask_from_lengths(src_lens, max_src_len)
    mel_pad = mask_from_lengths(mel_lens, max_mel_len) if mel_lens is not None else None
    emb = F.embedding(texts, sd["encoder.src_word_emb.weight"], padding_idx=0)
    x = emb + sd["encoder.position_enc"][:, : texts.shape[1]]
    enc = conformer_stack(sd, "encoder.", x, src_pad, c["encoder_layer"], c["encoder_head"], c["encoder_dropout"], training,
                          train_dropout, new_stats, taps)
    if taps is not None:
        taps["encoder_out"] = enc
    (x, p_targets, p_pred, e_targets, e_pred, log_d, d_rounded, mel_lens, mel_pad) = variance_adaptor(
        sd, model_cfg, pre_cfg, enc, src_lens, src_pad, mel_lens, mel_pad, max_mel_len, p_targets, e_targets, d_targets, None,
        p_control, e_control, d_control, train_dropout, taps)
    T = min(x.shape[1], model_cfg["max_seq_len"])          # conformer.py:148-154 crop (training and short inference)
    x = x[:, :T] + sd["decoder.position_enc"][:, :T]
    mel_pad = mel_pad[:, :T]
    dec = conformer_stack(sd, "decoder.", x, mel_pad, c["decoder_layer"], c["decoder_head"], c["decoder_dropout"], training,
                          train_dropout, new_stats, taps)
    if taps is not None:
        taps["decoder_out"] = dec
    mel = dec @ sd["mel_linear.weight"].t() + sd["mel_linear.bias"]
    post = postnet(sd, mel, training, train_dropout, new_stats) + mel
    return (mel, post, p_pred, e_pred, log_d, d_rounded, src_pad, mel_pad, src_lens, mel_lens, (None, None, None, None), None,
            p_targets, e_targets)
