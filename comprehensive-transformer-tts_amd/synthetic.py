"""Synthetic LJSpeech-shaped batches: no corpus is shipped with the
reference (only metadata + stats.json), so measurement and parity both run on seeded
synthetic inputs of the shapes dataset.py:166-227 would produce.

The positional layout of `as_model_args` is what train.py:106 passes: `model(*batch[2:])`.
"""
import torch

CANONICAL_SRC_LENS = [128, 123, 117, 115, 113, 112, 111, 96, 83, 82, 82, 81, 78, 63, 60, 55]
C1_SRC_LENS = [128, 113, 78, 60]


def _mel2ph_from_durations(d, src_lens):
    B, Ts = d.shape
    cum = torch.cumsum(d, 1)
    total = cum[:, -1]
    Tm = int(total.max())
    t = torch.arange(Tm)[None, :].expand(B, -1).contiguous()
    idx = torch.searchsorted(cum, t, right=True) + 1
    return torch.where(t < total[:, None], idx, torch.zeros_like(idx))


def make_batch(src_lens=None, frames_per_phone=8, seed=1234, n_mels=80, multi_speaker=False, speaker_dim=512,
               max_mel_cap=None):
    """Seeded CPU batch.  Returns a dict of CPU tensors; see `as_model_args`."""
    src_lens = list(CANONICAL_SRC_LENS if src_lens is None else src_lens)
    g = torch.Generator().manual_seed(seed)
    B = len(src_lens)
    src = torch.tensor(src_lens, dtype=torch.long)
    mel = src * frames_per_phone
    if max_mel_cap is not None:
        mel = torch.clamp(mel, max=max_mel_cap)
    Ts, Tm = int(src.max()), int(mel.max())
    texts = torch.randint(1, 360, (B, Ts), generator=g)
    texts[:, 5::6] = 357  # '@sp' so the word-duration loss is finite (loss.py:153-160)
    d = torch.zeros(B, Ts, dtype=torch.long)
    for b in range(B):
        n, extra = src_lens[b], int(mel[b]) - src_lens[b]
        d[b, :n] = 1
        if extra > 0:
            pick = torch.multinomial(torch.ones(n), extra, replacement=True, generator=g)
            d[b, :n] += torch.bincount(pick, minlength=n)
        texts[b, n:] = 0
    mels = torch.randn(B, Tm, n_mels, generator=g) * 1.5 - 5.0
    frame_valid = (torch.arange(Tm)[None, :] < mel[:, None])
    mels = mels * frame_valid[..., None]
    f0 = (torch.randn(B, Tm, generator=g) * 0.3 + 7.5) * frame_valid
    uv = (torch.rand(B, Tm, generator=g) < 0.3).float() * frame_valid
    cwt_spec = torch.randn(B, Tm, 10, generator=g) * frame_valid[..., None]
    energy = torch.randn(B, Ts, generator=g) * (torch.arange(Ts)[None, :] < src[:, None])
    batch = {
        "speakers": torch.zeros(B, dtype=torch.long),
        "texts": texts, "src_lens": src, "max_src_len": Ts,
        "mels": mels, "mel_lens": mel, "max_mel_len": Tm,
        "p_targets": {
            "pitch": f0.clone(), "f0": f0, "uv": uv, "cwt_spec": cwt_spec,
            "f0_mean": torch.full((B,), 5.3), "f0_std": torch.full((B,), 0.3),
            "mel2ph": _mel2ph_from_durations(d, src_lens),
        },
        "e_targets": energy, "d_targets": d,
        "spker_embeds": torch.randn(B, speaker_dim, generator=g) if multi_speaker else None,
    }
    return batch


def to_device(batch, device):
    out = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            out[k] = {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
        else:
            out[k] = v.to(device) if torch.is_tensor(v) else v
    return out


def make_unsup_batch(src_lens=None, frames_per_phone=8, seed=1234, **kw):
    """learn_alignment=True inputs: no duration targets, frame-level energy targets [B,Tm] and a positive
    attention prior [B,Ts,Tm] (a smooth diagonal band standing in for the beta-binomial prior of preprocessor.py:551-560)."""
    b = make_batch(src_lens, frames_per_phone, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 1)
    B, Ts, Tm = b["texts"].shape[0], b["texts"].shape[1], b["mels"].shape[1]
    P, M = b["src_lens"].float()[:, None, None], b["mel_lens"].float()[:, None, None]
    s_ = torch.arange(Ts)[None, :, None] / P
    t_ = torch.arange(Tm)[None, None, :] / M
    prior = torch.exp(-((t_ - s_) ** 2) / 0.02) + 0.05 * torch.rand(B, Ts, Tm, generator=g)
    valid = (torch.arange(Ts)[None, :, None] < P) & (torch.arange(Tm)[None, None, :] < M)
    b["attn_priors"] = prior * valid
    b["d_targets"] = None
    b["e_targets"] = torch.randn(B, Tm, generator=g) * (torch.arange(Tm)[None, :] < b["mel_lens"][:, None])
    return b


def as_model_args(batch):
    """Positional args in the order of CompTransTTS.forward (model/CompTransTTS.py:64-82)."""
    return (batch["speakers"], batch["texts"], batch["src_lens"], batch["max_src_len"], batch["mels"],
            batch["mel_lens"], batch["max_mel_len"], batch["p_targets"], batch["e_targets"], batch["d_targets"],
            batch.get("attn_priors"), batch["spker_embeds"])


def as_collated_tuple(batch):
    """The same batch in the layout `Dataset.collate_fn` hands to the train loop (dataset.py:166-228: a 20-tuple of numpy arrays and
    lists) - the input of `data.PackedBatch.pack`, i.e. of the host data path."""
    def n(t):
        return None if t is None else t.numpy()
    p = batch["p_targets"]
    B = batch["texts"].shape[0]
    return ([f"syn{i:04d}" for i in range(B)], [""] * B, n(batch["speakers"]), n(batch["texts"]), n(batch["src_lens"]),
            int(batch["max_src_len"]), n(batch["mels"]), n(batch["mel_lens"]), int(batch["max_mel_len"]), n(p["pitch"]), n(p["f0"]),
            n(p["uv"]), n(p["cwt_spec"]), n(p["f0_mean"]), n(p["f0_std"]), n(batch["e_targets"]), n(batch.get("d_targets")),
            n(p.get("mel2ph")) if batch.get("d_targets") is not None else None, n(batch.get("attn_priors")), n(batch.get("spker_embeds")))


def shard_indices(mel_lens, rank, world, order="strided"):
    """Which utterances of a global batch rank `rank` takes.
    "strided": r, r+world, ... - DistributedSampler order (train.py:44).  On a length-sorted batch (collate_fn sorts, dataset.py:230-236)
               rank 0 always gets the longest utterance of every group of `world`: at 16 utterances over 8 ranks that is 128+82 against
               96+55 phonemes - ~28 % more frames on rank 0 than on rank 7, and the step waits for rank 0.
    "snake":   utterances sorted by mel length, dealt 0..world-1, world-1..0, 0..: long and short ones pair up, same count per rank."""
    n = len(mel_lens)
    if order == "strided":
        return list(range(rank, n, world))
    if order != "snake":
        raise ValueError(f"shard order {order!r}: expected 'strided' or 'snake'")
    by_len = sorted(range(n), key=lambda i: (-int(mel_lens[i]), i))
    out = []
    for pos, i in enumerate(by_len):
        lap, k = divmod(pos, world)
        if (k if lap % 2 == 0 else world - 1 - k) == rank:
            out.append(i)
    return out


def shard_valid_frames(batch, world, order="strided"):
    """valid mel frames per rank of the sharded global batch -> list of `world` ints (the step time follows the maximum)"""
    return [int(sum(int(batch["mel_lens"][i]) for i in shard_indices(batch["mel_lens"], r, world, order))) for r in range(world)]


def shard(batch, rank, world, order="strided"):
    """Shard of a CPU batch (see shard_indices for the two orders); padded widths shrink to the shard's own maxima exactly as a
    per-rank `collate_fn` would produce them."""
    idx = shard_indices(batch["mel_lens"], rank, world, order)
    src, mel = batch["src_lens"][idx], batch["mel_lens"][idx]
    Ts, Tm = int(src.max()), int(mel.max())

    def cut(t, widths):
        if t is None:
            return None
        t = t[idx]
        for d, w in widths:
            t = t.narrow(d, 0, min(w, t.shape[d]))
        return t.contiguous()
    p = batch["p_targets"]
    frame_e = batch["e_targets"] is not None and batch["e_targets"].shape[1] == batch["mels"].shape[1] and batch.get("d_targets") is None
    out = {
        "speakers": batch["speakers"][idx], "texts": cut(batch["texts"], [(1, Ts)]), "src_lens": src, "max_src_len": Ts,
        "mels": cut(batch["mels"], [(1, Tm)]), "mel_lens": mel, "max_mel_len": Tm,
        "p_targets": {k: (cut(v, [(1, Tm)]) if v.dim() > 1 else v[idx]) for k, v in p.items()},
        "e_targets": cut(batch["e_targets"], [(1, Tm if frame_e else Ts)]), "d_targets": cut(batch.get("d_targets"), [(1, Ts)]),
        "spker_embeds": cut(batch.get("spker_embeds"), []),
    }
    if batch.get("attn_priors") is not None:
        out["attn_priors"] = cut(batch["attn_priors"], [(1, Ts), (2, Tm)])
    return out
