"""Where the torch glue launches of the xVAPitch C5 iteration come from: a TorchDispatchMode counts every aten operator that launches device work (views and
allocations excluded), forward and backward, by the innermost source line inside xva-trainer_amd.  python tools/c5_glue_map.py"""
import collections, os, runpy, sys, traceback
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
VIEWS = {"view", "reshape", "_unsafe_view", "slice", "as_strided", "transpose", "expand", "detach", "alias", "t", "permute", "unsqueeze", "squeeze", "select",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "unbind", "split", "split_with_sizes", "chunk", "narrow", "view_as", "_to_copy_",
         "lift_fresh", "record_stream", "is_pinned", "_local_scalar_dense", "unfold", "diagonal", "movedim", "flatten", "unflatten", "squeeze_", "unsqueeze_",
         "_reshape_alias", "size", "stride", "sym_size", "sym_stride", "sym_numel", "numel", "dim", "is_contiguous", "storage_offset", "sym_storage_offset"}
cnt = collections.Counter(); ops = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        out = func(*args, **(kwargs or {}))
        if name in VIEWS:
            return out
        ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        if ts and not any(t.is_cuda for t in ts) and not (isinstance(out, torch.Tensor) and out.is_cuda):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=14)):
            if "xva-trainer_amd" in fr.filename or "xva_trainer_amd" in fr.filename:
                site = "%s:%d" % (fr.filename.split("xvapitch/")[-1].split("amd/")[-1], fr.lineno)
                break
        cnt[(name, site)] += 1; ops[name] += 1
        return out
torch.cuda.synchronize()
with Mode():
    g["iteration"]()
torch.cuda.synchronize()
print("aten operators with device work in one iteration: %d" % sum(ops.values()))
print(dict(ops.most_common(30)))
site_tot = collections.Counter()
for (n, s), k in cnt.items():
    site_tot[s.split(":")[0]] += k
print(dict(site_tot.most_common()))
for (n, s), k in cnt.most_common(110):
    print("%5d  %-22s %s" % (k, n, s))
