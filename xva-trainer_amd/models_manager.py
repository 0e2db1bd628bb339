"""ModelsManager — the registry entry points server.py uses for the trainers (python/models_manager.py:115-128,152-163).

Only the trainer keys of the accelerated path are served ("fastpitch1_1", "hifigan"); the 16 dataset tools, the inference
wrappers and the xVAPitch trainer stay with the reference (`init_model` / `load_model` raise NotImplementedError here)."""
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
        else:
            raise NotImplementedError("trainer '%s' is not part of the accelerated path" % model_key)
        try:
            self.models_bank[model_key].model = self.models_bank[model_key].model
        except AttributeError:
            pass

    def load_model(self, model_key, ckpt_path, **kwargs):
        raise NotImplementedError("inference wrappers (infer_fastpitch1_1 / infer_hifigan) are a 'next' row (SURVEY.md §8f N4)")

    def set_device(self, device):
        if device == "cpu":
            raise RuntimeError("the MI355X-native training path has no CPU implementation")
        self.device_label = device
        self.device = torch.device(device)

    def models(self, key):
        return self.models_bank[key.lower()]
