import asyncio, logging, os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import data as D
from xva_trainer_amd.models_manager import ModelsManager
from xva_trainer_amd.data import SyntheticFastPitchLoader
tmp = tempfile.mkdtemp()
def mk(name, factory=None, compute="fp32"):
    mm = ModelsManager(logging.getLogger("t"), False, "cuda:0")
    mm.sync_init_model("fastpitch1_1", websocket=None, gpus=[0])
    tr = mm.models_bank["fastpitch1_1"]; tr.compute = compute; tr.loader_factory = factory; tr.init_logs(tmp + "/out/" + name)
    return tr
async def drive(tr, data, n):
    tr.running = True
    await tr.start.__func__(tr, data, gpus=[0]) if False else None
# scenario A: stage 2 synthetic
tr = mk("voice_o", lambda t: SyntheticFastPitchLoader(2, n_batches=4, t_text=10, t_mel=30, seed=1))
data = {"dataset_path": tmp + "/in/voice_o", "output_path": tmp + "/out", "checkpoint": None, "num_workers": 0, "batch_size": 4, "epochs_per_checkpoint": 1,
        "force_stage": 2, "max_iterations": 50001}
async def a():
    t0 = time.time()
    tr.running = False
    # replicate start() but bounded
    tr.running = True
    tr.force_stage = 2; tr.dataset_input = data["dataset_path"]; tr.dataset_id = "voice_o"; tr.dataset_output = tmp + "/out/voice_o"; tr.checkpoint = None
    tr.workers = 0; tr.batch_size = 4; tr.epochs_per_checkpoint = 1; tr.max_iterations = 50001; tr.synthetic_data = False
    tr.learning_rate, tr.weight_decay = 0.1, 1e-6; tr.dur_predictor_loss_scale = tr.pitch_predictor_loss_scale = 0.1; tr.attn_loss_scale = 1.0
    tr.warmup_steps, tr.grad_clip_thresh = 1000, 1000
    for i in range(12):
        if not tr.running: break
        await tr.iteration()
        print("A it", i, "total_iter", tr.total_iter, "acc", tr.accumulated_steps, "iter_loss", tr.iter_loss, "gam", tr.gam, "log", tr.training_log[-1:] , "%.2fs" % (time.time() - t0), flush=True)
asyncio.run(a())
# scenario B: stage 1 from files
ds = D.write_synthetic_dataset(tmp + "/in/voice_c", n_items=12, seed=4, min_s=0.5, max_s=1.0)
tr = mk("voice_c")
async def b():
    t0 = time.time()
    tr.running = True
    tr.force_stage = None; tr.dataset_input = ds; tr.dataset_id = "voice_c"; tr.dataset_output = tmp + "/out/voice_c"; tr.checkpoint = None
    tr.workers = 0; tr.batch_size = 1; tr.epochs_per_checkpoint = 1000; tr.max_iterations = 50006; tr.synthetic_data = False
    tr.learning_rate, tr.weight_decay = 0.1, 1e-6; tr.dur_predictor_loss_scale = tr.pitch_predictor_loss_scale = 0.1; tr.attn_loss_scale = 1.0
    tr.warmup_steps, tr.grad_clip_thresh = 1000, 1000
    for i in range(40):
        if not tr.running: break
        await tr.iteration()
        print("B it", i, "total_iter", tr.total_iter, "acc", tr.accumulated_steps, "iter_loss", tr.iter_loss, "gam", tr.gam, "B", tr.per_rank_batch, "len", len(tr.train_loader), tr.training_log[-1:], "%.2fs" % (time.time() - t0), flush=True)
asyncio.run(b())
