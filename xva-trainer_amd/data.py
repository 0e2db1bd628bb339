"""Batch sources for the trainers: the reference's dataset directory read from disk, batches built ON THE DEVICE.

The reference prepares batches on the CPU in DataLoader workers: `TTSDataset.__getitem__` / `get_mel` / `TTSCollate.__call__` /
`batch_to_gpu` (python/fastpitch1_1/fastpitch/data_function.py:297-352,385-429,565-741) and `MelDataset.__getitem__`
(python/hifigan/meldataset.py:340-373), with `.npy` side caches and sleep-and-retry races between workers.  Here the host only
reads files (stdlib `wave`, `np.load`) and concatenates the ragged items into pinned flat buffers; int16 -> float, peak
normalisation, random crop / zero pad, the TacotronSTFT mel, energy, the sort by text length, every zero-padding and the
beta-binomial prior run as HIP kernels (csrc/data_ops.hip, csrc/mel.hip: xva_data_*, xva_wav_*, xva_mel_spectrogram_ragged).

Dataset directory layout (the reference's): `metadata.csv` (`fname|text`), `wavs/*.wav` (22050 Hz mono int16), and the side
caches the reference writes next to it: `pitch/{name}.npy` (normalised f0 per mel frame; the reference fills it with
librosa.pyin, which is CPU preprocessing outside this path), `durs_text/{name}.npy` / `durs_arpabet/{name}.npy` (MAS durations
written after stage 1 — `FastPitchTrainer.extract_durations` here).  Text -> symbol ids uses the reference's `english_basic`
symbol table with a basic cleaner; the CMUdict / ARPAbet front end stays with the reference and plugs in as `text_encoder`.

The synthetic loaders at the bottom serve benchmarks and tests only (trainers use them only on an explicit opt-in).
"""
import ctypes as C
import os
import random
import re
import wave

import numpy as np
import torch

from . import _lib, synthetic
from .mel import TacotronSTFT

lib = _lib.lib
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
DATA_I16, DATA_I32, DATA_I64, DATA_F32, DATA_F64, DATA_F32_TRUNC = 0, 1, 2, 3, 4, 5
lib.xva_data_rank_desc.restype = i32
lib.xva_data_rank_desc.argtypes = [vp, i32, vp, vp]
lib.xva_data_pad_gather.restype = i32
lib.xva_data_pad_gather.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
lib.xva_mel_spectrogram_ragged.restype = i32
lib.xva_mel_spectrogram_ragged.argtypes = [C.POINTER(_lib.MelConfig), vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, i64, vp]
lib.xva_data_betabinom_prior.restype = i32
lib.xva_data_betabinom_prior.argtypes = [vp, vp, vp, i32, i32, i32, vp]
lib.xva_wav_peak_i16.restype = i32
lib.xva_wav_peak_i16.argtypes = [vp, vp, vp, i32, vp, vp]
lib.xva_wav_crop_norm.restype = i32
lib.xva_wav_crop_norm.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, C.c_double, i32, vp]

_TORCH_DT = {torch.int16: DATA_I16, torch.int32: DATA_I32, torch.int64: DATA_I64, torch.float32: DATA_F32, torch.float64: DATA_F64}


# ------------------------------------------------------------------------------------------------ files / text
def read_wav_int16(path):
    """load_wav_to_torch / load_wav (common/utils.py:42-48, hifigan/meldataset.py:57-60: scipy.io.wavfile.read): int16 PCM, mono."""
    with wave.open(path, "rb") as f:
        if f.getsampwidth() != 2 or f.getnchannels() != 1:
            raise ValueError("%s: expected 16-bit mono PCM (got %d-bit, %d channels)" % (path, 8 * f.getsampwidth(), f.getnchannels()))
        sr = f.getframerate()
        data = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
    return data, sr


def write_wav_int16(path, data, sr=22050):
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(np.asarray(data, dtype="<i2").tobytes())


_ARPABET = ["AA", "AA0", "AA1", "AA2", "AE", "AE0", "AE1", "AE2", "AH", "AH0", "AH1", "AH2", "AO", "AO0", "AO1", "AO2", "AW", "AW0", "AW1", "AW2",
            "AY", "AY0", "AY1", "AY2", "B", "CH", "D", "DH", "EH", "EH0", "EH1", "EH2", "ER", "ER0", "ER1", "ER2", "EY", "EY0", "EY1", "EY2", "F", "G",
            "HH", "IH", "IH0", "IH1", "IH2", "IY", "IY0", "IY1", "IY2", "JH", "K", "L", "M", "N", "NG", "OW", "OW0", "OW1", "OW2", "OY", "OY0", "OY1",
            "OY2", "P", "R", "S", "SH", "T", "TH", "UH", "UH0", "UH1", "UH2", "UW", "UW0", "UW1", "UW2", "V", "W", "Y", "Z", "ZH"]


class BasicTextEncoder:
    """`english_basic` symbol ids (common/text/symbols.py:15-21: '_' pad, '-', punctuation "!'(),.:;? ", A-Z a-z, 84 '@ARPAbet'
    = 148 symbols) behind a basic cleaner: lowercase, '/' -> ' ', collapsed whitespace, unknown characters dropped.  Number /
    abbreviation expansion and CMUdict ARPAbet substitution (`english_cleaners_v2`, p_arpabet) are the reference's text front
    end (out of the accelerated path); `{AA1 B}`-style ARPAbet groups already present in the text ARE encoded."""

    def __init__(self):
        self.symbols = list("_" + "-" + "!'(),.:;? " + "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz") + ["@" + s for s in _ARPABET]
        assert len(self.symbols) == synthetic.N_SYMBOLS
        self.to_id = {s: i for i, s in enumerate(self.symbols)}
        self.space = self.to_id[" "]
        self._curly = re.compile(r"(.*?)\{(.+?)\}(.*)")

    def _chars(self, text):
        text = re.sub(r"\s+", " ", text.lower().replace("/", " "))
        return [self.to_id[c] for c in text if c in self.to_id and c not in "_~"]

    def encode(self, text):
        out = []
        while text:
            m = self._curly.match(text)
            if not m:
                out += self._chars(text)
                break
            out += self._chars(m.group(1))
            out += [self.to_id["@" + p] for p in m.group(2).split() if "@" + p in self.to_id]
            text = m.group(3)
        return [self.space] + out + [self.space]           # TTSDataset.get_text: prepend / append the space symbol (:436-441)


def read_metadata(dataset_path, wav_dir="wavs"):
    """[(name_without_ext, wav_path, text)] for the lines of metadata.csv whose wav exists (common/utils.py:78-140); wav_dir: the folder of the
    clips (xVAPitch fine-tunes on `wavs_postprocessed/`, python/xvapitch/dataset.py:647)."""
    items = []
    with open(os.path.join(dataset_path, "metadata.csv"), encoding="utf-8") as f:
        for line in f.read().split("\n"):
            if not line.strip():
                continue
            parts = line.strip().split("|")
            fname = parts[0].split("/")[-1]
            if not fname.endswith(".wav"):
                fname += ".wav"
            path = os.path.join(dataset_path, wav_dir, fname)
            if os.path.exists(path):
                items.append((fname[:-4], path, parts[1] if len(parts) > 1 else ""))
    return items


# ------------------------------------------------------------------------------------------------ device collate
class _PinnedRing:
    """Persistent pinned staging buffers: a ring of RING buffers per dtype, grown when a batch needs more, instead of one
    cudaHostAlloc per field and batch.  A buffer is handed out again only after the H2D copy that read it last has run (its event)."""
    RING = 4

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._bufs = {}          # dtype -> [[tensor, event or None], ...]
        self._next = {}

    def take(self, n, dtype):
        if not torch.cuda.is_available():
            return torch.empty(n, dtype=dtype), None
        with self._lock:
            ring = self._bufs.setdefault(dtype, [[None, None] for _ in range(self.RING)])
            i = self._next.get(dtype, 0)
            self._next[dtype] = (i + 1) % self.RING
            slot = ring[i]
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < n:
            slot[0] = torch.empty(max(n, 1) * 5 // 4 + 16, dtype=dtype).pin_memory()
        slot[1] = torch.cuda.Event()
        return slot[0][:n], slot[1]


_PINNED = _PinnedRing()


def _walk_tensors(obj, fn, depth=0):
    if isinstance(obj, torch.Tensor):
        fn(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _walk_tensors(v, fn, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _walk_tensors(v, fn, depth + 1)
    elif depth < 3 and hasattr(obj, "__dict__"):
        for v in vars(obj).values():
            _walk_tensors(v, fn, depth + 1)


class Prefetcher:
    """One background thread keeps `depth` batches ahead of the training loop (the reference's DataLoader(num_workers, pin_memory,
    persistent_workers) — python/fastpitch1_1/xva_train.py:452, hifigan/xva_train.py:321): the file reads, the copy into the pinned ring, the H2D transfer
    and the device-side collate / mel kernels of batch i + 1 run on a side stream while step i runs on the trainer's stream.  The consumer makes its
    stream wait for the batch's event (no host synchronisation) and tells the caching allocator that its stream now uses the batch's tensors.
    Same batches in the same order as iterating `loader` directly (the loaders draw from their own seeded generators)."""
    _END = object()

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), int(depth)
        self._side = None

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):                      # actual_num_lines, items, ...: the wrapped loader's attributes
        return getattr(self.__dict__["loader"], name)

    def __iter__(self):
        import queue
        import threading
        if self.device.type != "cuda":
            yield from self.loader
            return
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        side, dev, END = self._side, self.device, self._END

        def put(x):
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                torch.cuda.set_device(dev)
                while not stop.is_set():
                    with torch.cuda.stream(side):
                        try:
                            b = next(it)
                        except StopIteration:
                            put(END)
                            return
                        ev = torch.cuda.Event()
                        ev.record(side)
                    if not put((b, ev)):
                        return
            except BaseException as e:                 # surfaces in the training thread at the next()
                put(e)

        th = threading.Thread(target=work, name="xva-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    return
                if isinstance(item, BaseException):
                    raise item
                b, ev = item
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                _walk_tensors(b, lambda t: t.record_stream(cur) if t.is_cuda else None)
                yield b
        finally:
            stop.set()
            try:
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=5.0)


class _Ragged:
    """Ragged host arrays -> one pinned flat buffer + offsets / lengths, copied to the device asynchronously."""

    def __init__(self, arrays, dtype, device, inner=1):
        lens = [a.shape[-1] for a in arrays]
        self.lens_host = lens
        off = np.zeros(len(arrays), dtype=np.int64)
        if len(arrays) > 1:
            off[1:] = np.cumsum(lens[:-1])
        total = int(sum(lens)) * inner
        flat, done = _PINNED.take(max(total, 1), dtype)
        np_flat = flat.numpy()
        pos = 0
        for a in arrays:
            n = a.size
            np_flat[pos:pos + n] = np.ascontiguousarray(a).reshape(-1)
            pos += n
        self.flat = flat.to(device, non_blocking=True)
        if done is not None:
            done.record()                 # the staging buffer may be refilled once this copy has run
        self.offsets = torch.from_numpy(off).to(device, non_blocking=True)
        self.lens = torch.tensor(lens, dtype=torch.int32).to(device, non_blocking=True)
        self.inner = inner


def pad_gather(rag, order, max_len, out_dtype, want_lens=False, trunc=False):
    """TTSCollate's zero padding of one ragged field, in `order` (data_function.py:574-660).  trunc: fp32 output holding trunc(x)
    (the reference accumulates pitch into a LongTensor, :594-606)."""
    B = rag.lens.numel()
    dst = torch.empty(B, rag.inner, max_len, device=rag.flat.device, dtype=out_dtype)
    lens_out = torch.empty(B, device=rag.flat.device, dtype=torch.int32) if want_lens else None
    _lib.check(lib.xva_data_pad_gather(_lib.ptr(rag.flat), _TORCH_DT[rag.flat.dtype], _lib.ptr(rag.offsets), _lib.ptr(rag.lens), _lib.ptr(order),
                                       _lib.ptr(dst), DATA_F32_TRUNC if trunc else _TORCH_DT[out_dtype], B, rag.inner, int(max_len), _lib.ptr(lens_out),
                                       _lib.stream_ptr()),
               "xva_data_pad_gather")
    return (dst, lens_out) if want_lens else dst


def rank_desc(lens):
    order = torch.empty_like(lens)
    _lib.check(lib.xva_data_rank_desc(_lib.ptr(lens), lens.numel(), _lib.ptr(order), _lib.stream_ptr()), "xva_data_rank_desc")
    return order


def betabinom_prior(in_lens, mel_lens, Tm, Tt):
    out = torch.empty(in_lens.numel(), Tm, Tt, device=in_lens.device, dtype=torch.float32)
    _lib.check(lib.xva_data_betabinom_prior(_lib.ptr(in_lens), _lib.ptr(mel_lens), _lib.ptr(out), in_lens.numel(), Tm, Tt, _lib.stream_ptr()),
               "xva_data_betabinom_prior")
    return out


class DeviceCollate:
    """TTSCollate + batch_to_gpu + the mel / energy half of TTSDataset.__getitem__ on the device (data_function.py:297-352,385-429,
    565-741).  `__call__(items, stage)` with items = [dict(wav=int16 array, text=int ids, pitch=(1, T) float | None,
    durs=float/int array | None)] returns an engine-ready fastpitch.engine.DeviceBatch."""

    def __init__(self, device, hop=256, n_fft=1024, reference_int_truncation=True):
        """reference_int_truncation: TTSCollate creates pitch_padded / energy_padded with the text's dtype (LongTensor,
        data_function.py:594-606), so the reference's batches carry pitch and energy truncated toward zero; True reproduces the
        reference batch bit for bit, False keeps the fractional values."""
        self.device = torch.device(device)
        self.trunc = bool(reference_int_truncation)
        self.stft = TacotronSTFT(n_fft, hop, n_fft, 80, 22050, 0.0, 8000.0).to(self.device)
        self.hop, self.n_fft = hop, n_fft
        self._ws = None

    def mel_ragged(self, wav_rag, order, n_max):
        eng = self.stft.engine
        B = wav_rag.lens.numel()
        T = eng.num_frames(n_max)
        need = int(lib.xva_mel_workspace_bytes(C.byref(eng.cfg), B, n_max))
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, device=self.device, dtype=torch.float32)
        mel = torch.empty(B, 80, T, device=self.device, dtype=torch.float32)
        energy = torch.empty(B, T, device=self.device, dtype=torch.float32)
        n_frames = torch.empty(B, device=self.device, dtype=torch.int32)
        _lib.check(lib.xva_mel_spectrogram_ragged(C.byref(eng.cfg), _lib.ptr(wav_rag.flat), _lib.ptr(wav_rag.offsets), _lib.ptr(wav_rag.lens),
                                                  _lib.ptr(order), B, int(n_max), _lib.ptr(eng.forward_basis), _lib.ptr(eng._mel_basis_padded),
                                                  _lib.ptr(mel), _lib.ptr(energy), int(self.trunc), _lib.ptr(n_frames), _lib.ptr(self._ws),
                                                  self._ws.numel() * 4, _lib.stream_ptr()), "xva_mel_spectrogram_ragged")
        return mel, energy, n_frames

    def __call__(self, items, stage):
        from .fastpitch.engine import DeviceBatch
        dev = self.device
        wav = _Ragged([it["wav"] for it in items], torch.int16, dev)
        text = _Ragged([np.asarray(it["text"], dtype=np.int32) for it in items], torch.int32, dev)
        order = rank_desc(text.lens)                                                    # sort by text length, descending (:569-572)
        n_max, t_max = max(wav.lens_host), max(text.lens_host)
        mel, energy, mel_lens = self.mel_ragged(wav, order, n_max)
        text_p, in_lens = pad_gather(text, order, t_max, torch.int32, want_lens=True)
        Tm = mel.size(2)
        pitch = durs = prior = None
        if stage in (3, 4, -1):                                                          # pitch / energy only reach the batch in these stages (:318-331,592-610)
            pr = _Ragged([np.asarray(it["pitch"], dtype=np.float32).reshape(1, -1) for it in items], torch.float32, dev)
            pitch = pad_gather(pr, order, Tm, torch.float32, trunc=self.trunc)
        else:
            energy = None
        if stage not in (1, -1):
            dr = _Ragged([np.asarray(it["durs"], dtype=np.float32) for it in items], torch.float32, dev)
            durs = pad_gather(dr, order, t_max, torch.int32).squeeze(1)        # durs_padded is a LongTensor: float durations truncate (:612-636)
        else:
            prior = betabinom_prior(in_lens, mel_lens, Tm, t_max)
        b = DeviceBatch(text_p.squeeze(1), in_lens, mel, mel_lens, pitch, energy, durs)
        b.attn_prior = prior
        b.order = order
        return b


class _BoundedCache(dict):
    """Decoded clips kept in host memory up to `cap` items, oldest first out (the reference caps its caches the same way: 3000 items in
    TTSDataset, 5000 in MelDataset — multi-hour datasets times N ranks must not exhaust host RAM)."""

    def __init__(self, cap):
        super().__init__()
        self.cap = int(cap)

    def __setitem__(self, k, v):
        if k not in self and len(self) >= self.cap:
            del self[next(iter(self))]
        super().__setitem__(k, v)


# ------------------------------------------------------------------------------------------------ FastPitch loader
class FastPitchFileLoader:
    """DataLoader(TTSDataset, TTSCollate, shuffle=True, drop_last=True) (python/fastpitch1_1/xva_train.py:437-452) with the batch
    built on the device.  Data-parallel: every rank shuffles the same epoch order and takes a disjoint stride of it."""

    def __init__(self, dataset_path, batch_size, stage, device, text_encoder=None, seed=1234, rank=0, world=1, dm=1, shuffle=True,
                 durs_kind="text"):
        self.path, self.batch_size, self.stage, self.device = dataset_path, int(batch_size), int(stage), torch.device(device)
        self.enc = text_encoder or BasicTextEncoder()
        self.items = read_metadata(dataset_path)
        if not self.items:
            raise FileNotFoundError("no usable lines in %s/metadata.csv (wavs/ missing?)" % dataset_path)
        self.actual_num_lines = len(self.items)
        self.index = list(range(len(self.items))) * max(1, int(dm))                    # load_filepaths_and_text's dm repetition
        self.seed, self.rank, self.world, self.shuffle, self.epoch = seed, rank, world, shuffle, 0
        self.durs_kind = durs_kind
        self.collate = DeviceCollate(self.device)
        self._cache = _BoundedCache(3000)

    def __len__(self):
        return (len(self.index) // self.world) // self.batch_size

    def item(self, i):
        it = self._cache.get(i)
        if it is None:
            name, path, text = self.items[i]
            wav, sr = read_wav_int16(path)
            if sr != 22050:
                raise ValueError("%s SR doesn't match target 22050 SR" % path)
            it = {"name": name, "path": path, "wav": wav, "text": np.asarray(self.enc.encode(text), dtype=np.int32)}
            n_frames = 1 + wav.shape[0] // 256
            if self.stage in (3, 4, -1):
                ppath = os.path.join(self.path, "pitch", name + ".npy")
                if not os.path.exists(ppath):
                    raise FileNotFoundError("%s is missing: the pitch cache is written by the reference's preprocessing (librosa.pyin, "
                                            "data_function.py:525-560), which is outside the accelerated path" % ppath)
                p = np.load(ppath).astype(np.float32).reshape(1, -1)
                if p.shape[1] != n_frames:
                    raise ValueError("%s: %d pitch frames for %d mel frames" % (ppath, p.shape[1], n_frames))
                it["pitch"] = p
            if self.stage not in (1, -1):
                dpath = os.path.join(self.path, "durs_" + self.durs_kind, name + ".npy")
                if not os.path.exists(dpath):
                    raise FileNotFoundError("%s is missing: durations are extracted after training stage 1 (extract_durations)" % dpath)
                it["durs"] = np.load(dpath).astype(np.float32).reshape(-1)
            self._cache[i] = it
        return it

    def __iter__(self):
        order = list(self.index)
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        self.epoch += 1
        order = order[self.rank::self.world]
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield self.collate([self.item(i) for i in idx], self.stage)


# ------------------------------------------------------------------------------------------------ HiFi-GAN loader
def prepare_segments(clips, starts, segment, device, normalize=True):
    """MelDataset.__getitem__'s audio path (meldataset.py:345-361) for a batch of int16 clips on the device:
    x / 32768 -> librosa.util.normalize (peak) * 0.95 -> [start, start + segment) or right zero pad.  Returns (B, segment) fp32."""
    rag = _Ragged([np.asarray(c, dtype=np.int16) for c in clips], torch.int16, device)
    B = len(clips)
    peak = torch.empty(B, device=rag.flat.device, dtype=torch.int32)
    _lib.check(lib.xva_wav_peak_i16(_lib.ptr(rag.flat), _lib.ptr(rag.offsets), _lib.ptr(rag.lens), B, _lib.ptr(peak), _lib.stream_ptr()), "xva_wav_peak_i16")
    st = torch.tensor(list(starts), dtype=torch.int32).to(rag.flat.device, non_blocking=True)
    out = torch.empty(B, segment, device=rag.flat.device, dtype=torch.float32)
    _lib.check(lib.xva_wav_crop_norm(_lib.ptr(rag.flat), _lib.ptr(rag.offsets), _lib.ptr(rag.lens), _lib.ptr(st), _lib.ptr(peak), _lib.ptr(out), B,
                                     int(segment), 0.95, int(bool(normalize)), _lib.stream_ptr()), "xva_wav_crop_norm")
    return out


class HifiFileLoader:
    """DataLoader(MelDataset(get_dataset_filelist(...)), shuffle, drop_last) (python/hifigan/xva_train.py:309-325, meldataset.py:268-373):
    yields (B, segment) fp32 waveform crops on the device; the two mels per item are computed by the trainer with the HIP mel."""

    def __init__(self, dataset_path, batch_size, device, segment=8192, seed=1234, rank=0, world=1, dm=None):
        files = [p for _, p, _ in read_metadata(dataset_path)]
        if not files:
            raise FileNotFoundError("no usable lines in %s/metadata.csv (wavs/ missing?)" % dataset_path)
        if dm is None:
            dm = max(1, round(1000 / len(files)))                                       # get_dataset_filelist (:298-300)
        rng = random.Random(seed)
        self.files = []
        for _ in range(dm):
            rng.shuffle(files)
            self.files += files
        self.batch_size, self.segment, self.device, self.rank, self.world = int(batch_size), int(segment), torch.device(device), rank, world
        # The epoch shuffle must be IDENTICAL on every rank (order[rank::world] is then a disjoint shard): Random(seed + epoch) like
        # FastPitchFileLoader.  The crop position of an item is a function of (seed, epoch, the item's place in the GLOBAL epoch order), drawn from a
        # second stream: the union of the ranks' batches is the batch a single process with world x the batch size would build, crops included.
        self.seed, self.epoch = seed, 0
        self._cache = _BoundedCache(5000)

    def __len__(self):
        return (len(self.files) // self.world) // self.batch_size

    def clip(self, path):
        c = self._cache.get(path)
        if c is None:
            c, sr = read_wav_int16(path)
            self._cache[path] = c
        return c

    def __iter__(self):
        order = list(self.files)
        random.Random(self.seed + 1 + self.epoch).shuffle(order)
        crop_rng = random.Random(self.seed * 7919 + 1 + self.epoch)
        crop_u = [crop_rng.random() for _ in order]
        self.epoch += 1
        order, crop_u = order[self.rank::self.world], crop_u[self.rank::self.world]
        for b in range(len(self)):
            sl = slice(b * self.batch_size, (b + 1) * self.batch_size)
            clips = [self.clip(p) for p in order[sl]]
            # random.randint(0, len - segment) (meldataset.py:354-358), from the item's own uniform draw
            starts = [min(len(c) - self.segment, int(u * (len(c) - self.segment + 1))) if len(c) >= self.segment else 0 for c, u in zip(clips, crop_u[sl])]
            yield prepare_segments(clips, starts, self.segment, self.device)


# ------------------------------------------------------------------------------------------------ synthetic (bench / tests)
def beta_binomial_prior_distribution(phoneme_count, mel_count, scaling=1.0):
    """The (mel_count, phoneme_count) beta-binomial attention prior of one utterance — same call and formula as
    python/fastpitch1_1/fastpitch/data_function.py:84-94 (scipy.stats.betabinom).  Host twin of xva_data_betabinom_prior."""
    from scipy.stats import betabinom
    x = np.arange(0, phoneme_count)
    rows = [betabinom(phoneme_count, scaling * i, scaling * (mel_count + 1 - i)).pmf(x) for i in range(1, mel_count + 1)]
    return torch.tensor(np.array(rows))


def collate_attn_prior(in_lens, mel_lens):
    """TTSCollate's zero-padded (B, max_mel, max_text) stack of the per-item priors (data_function.py:600-609)."""
    out = torch.zeros(len(in_lens), int(max(mel_lens)), int(max(in_lens)))
    for b, (L, M) in enumerate(zip(in_lens, mel_lens)):
        out[b, :int(M), :int(L)] = beta_binomial_prior_distribution(int(L), int(M)).float()
    return out


class SyntheticFastPitchLoader:
    """Yields (x, y, num_frames)-ready dict batches shaped like TTSCollate's output (data_function.py:565-695).
    with_prior adds the beta-binomial `attn_prior` training stage 1 consumes."""

    def __init__(self, batch_size, n_batches=8, t_text=150, t_mel=860, seed=1234, ragged=True, with_prior=False):
        self.batches = [synthetic.fastpitch_batch(batch_size, t_text, t_mel, seed + i, ragged=ragged) for i in range(n_batches)]
        if with_prior:
            for b in self.batches:
                b["attn_prior"] = collate_attn_prior(b["in_lens"].tolist(), b["mel_lens"].tolist())

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


class SyntheticHifiLoader:
    """Yields (wav (B, segment) fp32 in [-1, 1]) crops like MelDataset.__getitem__ (meldataset.py:340-373); the mels
    are computed on the GPU by the trainer (the reference computes them on the CPU in the dataset)."""

    def __init__(self, batch_size, n_batches=8, segment=8192, seed=4321):
        self.items = []
        for i in range(n_batches):
            wav = np.stack([synthetic.synth_wave(segment, seed + i * batch_size + j) for j in range(batch_size)])
            wav = wav / np.abs(wav).max(axis=1, keepdims=True) * 0.95
            self.items.append(torch.from_numpy(wav.astype(np.float32)))

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def write_synthetic_dataset(path, n_items=8, seed=0, min_s=1.0, max_s=3.0, with_pitch=True, sr=22050, with_se_embs=False, min_words=2, fixed_text=None):
    """A reference-layout dataset directory of synthetic clips (tests / smoke): metadata.csv, wavs/*.wav (int16) and, with_pitch,
    the `pitch/*.npy` cache in the reference's format ((1, n_frames) float32, zeros = unvoiced; data_function.py:525-560).  min_words: the shortest
    line (xVAPitch drops lines under 15 characters); fixed_text: one text for every clip (equal token counts)."""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(path, "wavs"), exist_ok=True)
    if with_pitch:
        os.makedirs(os.path.join(path, "pitch"), exist_ok=True)
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima"]
    lines = []
    for i in range(n_items):
        n = int(rng.uniform(min_s, max_s) * sr)
        wav = np.round(synthetic.synth_wave(n, seed * 1000 + i) * 32768.0).astype(np.int16)
        name = "clip_%04d" % i
        write_wav_int16(os.path.join(path, "wavs", name + ".wav"), wav, sr)
        text = " ".join(words[int(k)] for k in rng.randint(0, len(words), size=min_words + i % 5)) + "."
        if fixed_text is not None:
            text = fixed_text
        lines.append("%s|%s" % (name, text))
        if with_pitch:
            T = 1 + n // 256
            p = rng.randn(1, T).astype(np.float32)
            p[0, rng.rand(T) < 0.3] = 0.0
            np.save(os.path.join(path, "pitch", name + ".npy"), p)
        if with_se_embs:                  # xVAPitch: the 512-d speaker embedding of each clip (python/xvapitch/get_dataset_emb.py reads se_embs/*.npy)
            os.makedirs(os.path.join(path, "se_embs"), exist_ok=True)
            np.save(os.path.join(path, "se_embs", name + ".npy"), (rng.randn(512) * 0.05 + np.linspace(-1, 1, 512)).astype(np.float32))
    with open(os.path.join(path, "metadata.csv"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    return path
