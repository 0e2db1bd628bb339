"""ModelsManager — the registry entry points server.py uses for the trainers (python/models_manager.py:115-128,152-163).

Served here: the trainer keys of the accelerated path ("fastpitch1_1", "hifigan", "xvapitch": python/models_manager.py:105-128) and the
FastPitch / HiFi-GAN / xVAPitch inference wrappers ("infer_fastpitch1_1", "infer_hifigan", "infer_xvapitch", :130-150); the 16 dataset tools stay
with the reference (`init_model` raises NotImplementedError for them here)."""
import os

import torch


class ModelsManager(object):
    def __init__(self, logger, PROD, device="cpu"):
        self.models_bank = {}
        self.logger = logger
        self.PROD = PROD
        self.device_label = device
        self.device = torch.device(device)

    async def init_model(self, model_key, websocket=None, gpus=[0]):
        raise NotImplementedError("dataset tools / inference models are served by the reference's models_manager (out of scope)")

    def sync_init_model(self, model_key, websocket=None, gpus=[0]):
        model_key = model_key.lower()
        if model_key in self.models_bank and self.models_bank[model_key] != 0:
            return
        if model_key == "fastpitch1_1":
            from .fastpitch.xva_train import FastPitchTrainer
            self.models_bank[model_key] = FastPitchTrainer(self.logger, self.PROD, gpus, self, websocket=websocket)
        elif model_key == "hifigan":
            from .hifigan.xva_train import HiFiTrainer
            self.models_bank[model_key] = HiFiTrainer(self.logger, self.PROD, gpus, self, websocket=websocket)
        elif model_key == "xvapitch":
            from .xvapitch.xva_train import xVAPitchTrainer
            self.models_bank[model_key] = xVAPitchTrainer(self.logger, self.PROD, gpus, self, websocket=websocket)
        else:
            raise NotImplementedError("trainer '%s' is not part of the accelerated path" % model_key)
        try:
            self.models_bank[model_key].model = self.models_bank[model_key].model
        except AttributeError:
            pass

    def load_model(self, model_key, ckpt_path, **kwargs):
        """python/models_manager.py:130-150: lazily build the inference wrapper, return "ENOENT" for a missing file, (re)load the
        checkpoint when the path changed."""
        if model_key not in self.models_bank:
            dev = self.device if self.device.type == "cuda" else torch.device("cuda", 0)
            if model_key == "infer_fastpitch1_1":
                from .infer import FastPitch1_1
                self.models_bank[model_key] = FastPitch1_1(self.logger, self.PROD, dev, self)
            elif model_key == "infer_hifigan":
                from .infer import HiFi_GAN
                self.models_bank[model_key] = HiFi_GAN(self.logger, self.PROD, dev, self)
            elif model_key == "infer_xvapitch":
                from .xvapitch.xva_train import xVAPitchModel
                self.models_bank[model_key] = xVAPitchModel(self.logger, self.PROD, dev, self)
            else:
                raise NotImplementedError("inference model '%s' is not part of the accelerated path" % model_key)
        if not os.path.exists(ckpt_path):
            return "ENOENT"
        if self.models_bank[model_key].ckpt_path != ckpt_path:
            ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            self.models_bank[model_key].load_state_dict(ckpt_path, ckpt, **kwargs)

    def set_device(self, device):
        """python/models_manager.py:152-161 ("gpu" -> "cuda"); the MI355X-native path has no CPU implementation."""
        if device == "gpu":
            device = "cuda"
        if device == "cpu":
            raise RuntimeError("the MI355X-native path has no CPU implementation")
        if self.device_label == device:
            return
        self.device_label = device
        self.device = torch.device(device)
        for key, m in list(self.models_bank.items()):
            if hasattr(m, "set_device"):
                m.set_device(self.device)

    def models(self, key):
        return self.models_bank[key.lower()]
