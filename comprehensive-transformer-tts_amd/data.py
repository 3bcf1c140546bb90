"""Host data path of the train step (reference dataset.py:166-248 `Dataset.reprocess/collate_fn`,
utils/tools.py:69-147 `to_device`).

The reference pads every field with numpy per batch, converts dtypes on the host and issues ~16 separate pageable H2D copies
(`torch.from_numpy(..).long().to(device)`), all on the training thread with `num_workers=0`.  Once a train step is ~30 ms that is the
bottleneck.  Here:

  * `collate(samples, ...)`      - same grouping / padding rules and the same 20-tuple as `Dataset.collate_fn` (bit-identical arrays);
  * `PackedBatch.pack(tuple)`    - every array is written ONCE, already in its device dtype (int64 / float32), into one pinned host
                                   buffer with 256-byte aligned segments;
  * `PackedBatch.to_device(..)`  - ONE asynchronous H2D copy on a side stream; the model inputs are zero-copy views of the device
                                   buffer, returned in the layout of the reference's `to_device` (14-list with the pitch dict);
  * `Prefetcher`                 - a background thread that collates, packs and uploads `depth` batches ahead; the consumer only
                                   waits on an event.

No arithmetic happens here; the module is host plumbing (numpy + pinned memory + streams).
"""
import queue
import threading

import numpy as np
import torch

_ALIGN = 256


def _pad_1d(xs, pad=0):
    n = max(len(x) for x in xs)
    out = np.full((len(xs), n), pad, dtype=np.result_type(*[np.asarray(x).dtype for x in xs]))
    for i, x in enumerate(xs):
        out[i, :len(x)] = x
    return out


def _pad_2d(xs):
    n = max(np.shape(x)[0] for x in xs)
    out = np.zeros((len(xs), n, np.shape(xs[0])[1]), dtype=np.result_type(*[np.asarray(x).dtype for x in xs]))
    for i, x in enumerate(xs):
        out[i, :np.shape(x)[0]] = x
    return out


def reprocess(data, idxs, learn_alignment=False, pitch_cwt=True, load_spker_embed=False):
    """`Dataset.reprocess` (dataset.py:166-228): gather the samples `idxs` of `data` and pad each field to the batch maximum."""
    g = lambda k: [data[i][k] for i in idxs]                                     # noqa: E731
    texts, mels = g("text"), g("mel")
    text_lens = np.array([t.shape[0] for t in texts])
    mel_lens = np.array([m.shape[0] for m in mels])
    cwt_specs = f0_means = f0_stds = None
    if pitch_cwt:
        cwt_specs = _pad_2d(g("cwt_spec"))
        f0_means, f0_stds = np.array(g("f0_mean")), np.array(g("f0_std"))
    durations = mel2phs = attn_priors = None
    if learn_alignment:
        attn_priors = np.zeros((len(idxs), int(text_lens.max()), int(mel_lens.max())), dtype=np.float32)   # pad_3D, tools.py:570-574
        for i, a in enumerate(g("attn_prior")):
            attn_priors[i, :a.shape[0], :a.shape[1]] = a
    else:
        durations, mel2phs = _pad_1d(g("duration")), _pad_1d(g("mel2ph"))
    spk = np.concatenate(np.array(g("spker_embed")), axis=0) if load_spker_embed else None
    return (g("id"), g("raw_text"), np.array(g("speaker")), _pad_1d(texts), text_lens, max(text_lens), _pad_2d(mels), mel_lens,
            max(mel_lens), _pad_1d(g("pitch")), _pad_1d(g("f0")), _pad_1d(g("uv")), cwt_specs, f0_means, f0_stds, _pad_1d(g("energy")),
            durations, mel2phs, attn_priors, spk)


def collate(data, batch_size, sort=False, drop_last=False, **kw):
    """`Dataset.collate_fn` (dataset.py:230-248): optional sort by descending text length, chunks of `batch_size`, the tail kept as
    a short batch unless `drop_last`."""
    n = len(data)
    idx = np.argsort(-np.array([d["text"].shape[0] for d in data])) if sort else np.arange(n)
    tail = idx[n - (n % batch_size):]
    idx = idx[:n - (n % batch_size)].reshape((-1, batch_size)).tolist()
    if not drop_last and len(tail) > 0:
        idx += [tail.tolist()]
    return [reprocess(data, ix, **kw) for ix in idx]


# field -> (position in the 20-tuple, device dtype the reference's to_device produces; None = keep the numpy dtype)
_FIELDS = [("speakers", 2, np.int64), ("texts", 3, np.int64), ("src_lens", 4, None), ("mels", 6, np.float32), ("mel_lens", 7, None),
           ("pitches", 9, np.int64), ("f0s", 10, np.float32), ("uvs", 11, np.float32), ("cwt_specs", 12, np.float32),
           ("f0_means", 13, np.float32), ("f0_stds", 14, np.float32), ("energies", 15, None), ("durations", 16, np.int64),
           ("mel2phs", 17, np.int64), ("attn_priors", 18, np.float32), ("spker_embeds", 19, np.float32)]


class PackedBatch:
    """One collated batch as a single pinned host buffer + layout table."""

    def __init__(self, ids, raw_texts, max_src_len, max_mel_len, layout, host):
        self.ids, self.raw_texts = ids, raw_texts
        self.max_src_len, self.max_mel_len = int(max_src_len), int(max_mel_len)
        self.layout = layout                # name -> (byte offset, shape, torch dtype)
        self.host = host                    # uint8 tensor (pinned when CUDA is available)

    @staticmethod
    def pack(batch, pin=None):
        arrays, off = {}, 0
        for name, pos, dt in _FIELDS:
            a = batch[pos]
            if a is None:
                continue
            a = np.ascontiguousarray(a if dt is None else np.asarray(a).astype(dt, copy=False))
            arrays[name] = (off, a)
            off += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        pin = torch.cuda.is_available() if pin is None else pin
        host = torch.empty(max(off, 1), dtype=torch.uint8, pin_memory=pin)
        hv = host.numpy()
        layout = {}
        for name, (o, a) in arrays.items():
            hv[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
            layout[name] = (o, tuple(a.shape), torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype)
        return PackedBatch(batch[0], batch[1], batch[5], batch[8], layout, host)

    def _views(self, buf):
        out = {}
        for name, (o, shape, dtype) in self.layout.items():
            n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            out[name] = buf[o:o + n].view(dtype).view(shape)
        return out

    def as_reference_list(self, views):
        """the 14-list of `utils/tools.py:69-134 to_device` (pitch fields gathered in a dict)"""
        v = views.get
        pitch = {"pitch": v("pitches"), "f0": v("f0s"), "uv": v("uvs"), "cwt_spec": v("cwt_specs"), "f0_mean": v("f0_means"),
                 "f0_std": v("f0_stds"), "mel2ph": v("mel2phs")}
        return [self.ids, self.raw_texts, v("speakers"), v("texts"), v("src_lens"), self.max_src_len, v("mels"), v("mel_lens"),
                self.max_mel_len, pitch, v("energies"), v("durations"), v("attn_priors"), v("spker_embeds")]

    def host_views(self):
        return self.as_reference_list(self._views(self.host))

    def to_device(self, device, stream=None):
        """ONE async H2D copy; returns (reference-style 14-list of device views, event recorded after the copy)."""
        if torch.device(device).type != "cuda":
            raise RuntimeError("PackedBatch.to_device targets the HIP device; use host_views() to inspect a batch on the host")
        stream = stream or torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            dev = torch.empty(self.host.numel(), dtype=torch.uint8, device=device)
            dev.copy_(self.host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        self.device_buffer = dev              # consumers on another stream must dev.record_stream(their stream) (Prefetcher does)
        return self.as_reference_list(self._views(dev)), ev


class Prefetcher:
    """Background collate + pack + upload, `depth` batches ahead of the training loop.
    `batches`: iterable of 20-tuples (e.g. the lists `collate` returns).  Iterating yields reference-style 14-lists whose tensors
    are already resident; the consumer stream is made to wait on the copy event (no host sync)."""

    def __init__(self, batches, device, depth=2):
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.q = queue.Queue(maxsize=depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.err = None
        self.t = threading.Thread(target=self._run, args=(iter(batches),), daemon=True)
        self.t.start()

    def _run(self, it):
        try:
            torch.cuda.set_device(self.device)
            for b in it:
                pb = PackedBatch.pack(b)
                views, ev = pb.to_device(self.device, self.copy_stream)
                self.q.put((views, ev, pb.device_buffer))
        except Exception as e:          # noqa: BLE001   (surface worker errors in the consumer)
            self.err = e
        self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            if self.err is not None:
                raise self.err
            raise StopIteration
        batch, ev, dev = item
        self.last_buffer = dev            # the packed device buffer behind `batch` (trainer.TrainStep.feed copies it in one piece)
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        dev.record_stream(cur)            # the buffer was allocated on the copy stream: keep the allocator from recycling it early
        return batch
