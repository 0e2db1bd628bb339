"""Inference wrappers behind ModelsManager.load_model — `FastPitch1_1` (python/fastpitch1_1/xva_train.py:1172-1233) and `HiFi_GAN`
(python/hifigan/models.py:301-331) on the HIP engines.  Same constructor signatures, `load_state_dict(ckpt_path, ckpt)`,
`set_device`, `isReady` / `ckpt_path` / `model` attributes and `infer(...)` that writes a 22050 Hz int16 wav, so the UI's
preview / export path (server.py:313-330) keeps working against checkpoints written by either trainer."""
import json
import re

import numpy as np
import torch

from . import _lib
from .data import BasicTextEncoder, write_wav_int16
from .fastpitch.model import FastPitch
from .hifigan import engine as HE


class Generator:
    """`hifigan.models.Generator` as the wrappers use it: `model(mel) -> (B, 1, T * 256)`, `load_state_dict(sd)`, `state_dict()`, `to()`."""

    def __init__(self, device, compute="bf16"):
        self.device = torch.device(device)
        self.eng = HE.HifiganEngine(self.device, compute)
        self.flat = torch.zeros(self.eng.total[HE.G], device=self.device)

    def load_state_dict(self, sd, strict=True):
        HE.to_flat(sd, self.eng.table[HE.G], self.flat)

    def state_dict(self):
        return HE.from_flat(self.flat, self.eng.table[HE.G])

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.XvaError("the MI355X-native path has no CPU implementation")
        if device != self.device:
            flat = self.flat.to(device)
            self.device, self.eng = device, HE.HifiganEngine(device, "bf16" if self.eng.dt else "fp32")
            self.flat = flat
        return self

    def eval(self):
        return self

    def __call__(self, mel):
        """mel (B, 80, T) -> waveform (B, 1, T * 256) (models.py:110-128), any T >= 1."""
        with torch.no_grad():
            return self.eng.generator_forward(self.flat, mel.to(self.device)).unsqueeze(1)


class HiFi_GAN(object):
    def __init__(self, logger, PROD, device, models_manager, config_file=None):
        self.logger, self.PROD, self.models_manager = logger, PROD, models_manager
        self.device = torch.device(device)
        self.ckpt_path = None
        self.model = Generator(self.device)
        self.isReady = True

    def load_state_dict(self, ckpt_path, sd):
        self.ckpt_path = ckpt_path
        self.model.load_state_dict(sd["generator"])

    def set_device(self, device):
        self.device = torch.device(device)
        self.model = self.model.to(self.device)


class FastPitch1_1(object):
    def __init__(self, logger, PROD, device, models_manager):
        self.logger, self.PROD, self.models_manager = logger, PROD, models_manager
        self.device = torch.device(device)
        self.ckpt_path = None
        self.arpabet_dict = {}
        self.text_encoder = BasicTextEncoder()
        self.init_model("english_basic")
        self.isReady = True

    def init_model(self, symbols_alphabet):
        if symbols_alphabet != "english_basic":
            raise NotImplementedError("symbols_alphabet %r: only english_basic (148 symbols) is built" % symbols_alphabet)
        self.symbols_alphabet = symbols_alphabet
        self.model = FastPitch(logger=self.logger).to(self.device)
        self.model.eval()
        self.model.device = self.device

    def load_state_dict(self, ckpt_path, ckpt, n_speakers=1):
        self.ckpt_path = ckpt_path
        try:
            with open(ckpt_path.replace(".pt", ".json"), "r") as f:
                data = json.load(f)
            if "symbols_alphabet" in data and data["symbols_alphabet"] != self.symbols_alphabet:
                self.init_model(data["symbols_alphabet"])
        except FileNotFoundError:
            pass
        if "state_dict" in ckpt:
            ckpt = ckpt["state_dict"]
        self.model.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in ckpt.items()}, strict=False)
        self.model.eval()

    def set_device(self, device):
        self.device = torch.device(device)
        self.model = self.model.to(self.device)
        self.model.device = self.device

    def infer(self, plugin_manager, text, output, vocoder, speaker_i, pace=1.0, pitch_data=None, old_sequence=None, globalAmplitudeModifier=None):
        """xva_train.py:1212-1233: text -> symbol ids -> FastPitch.infer -> HiFi-GAN generator -> int16 wav file."""
        text = re.sub(r"[^a-zA-ZäöüÄÖÜß\s\(\)\[\]0-9\?\.\,\!\'\{\}]+", "", text)
        text = text.replace("(", "").replace(")", "")
        ids = self.text_encoder.encode(text)[1:-1]                      # text_to_sequence adds no surrounding spaces
        seq = torch.LongTensor(ids).unsqueeze(0).to(self.device)
        with torch.no_grad():
            mel, mel_lens, _, _, _ = self.model.infer(seq, pace=pace)
            y_g_hat = self.models_manager.models("infer_hifigan").model(mel)
            audio = (y_g_hat.squeeze() * 32768.0).cpu().numpy().astype("int16")
        write_wav_int16(output, audio, 22050)
        return ""
