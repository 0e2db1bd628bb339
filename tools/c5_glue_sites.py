"""Attribute the torch-native device work of one xVAPitch iteration to the package lines that issue it.

    XVA_C5_GLUE_SITES=1 python tools/c5_step_time.py ...   -> gpurun_out/c5_trace.json.gz (torch.profiler chrome trace with Python frames)
    python tools/c5_glue_sites.py gpurun_out/c5_trace.json.gz

Every CPU-side aten op that launches device work (an at::native kernel, a hipMemcpy* / hipMemset* runtime call) is assigned to the innermost
`python_function` event of the package that encloses it on the same thread; ops issued from the autograd thread without an enclosing package frame
(AccumulateGrad, gradient fan-in adds) are listed by their aten name."""
import bisect
import collections
import gzip
import json
import sys

path = sys.argv[1]
op = gzip.open if path.endswith(".gz") else open
with op(path, "rt") as f:
    tr = json.load(f)
evs = [e for e in tr["traceEvents"] if e.get("ph") == "X"]
by_tid_py = collections.defaultdict(list)
runtime, ops, kernels_by_corr = [], [], {}
for e in evs:
    cat = e.get("cat", "")
    if cat == "python_function":
        by_tid_py[(e["pid"], e["tid"])].append(e)
    elif cat in ("cuda_runtime", "cuda_driver"):
        runtime.append(e)
    elif cat == "cpu_op":
        ops.append(e)
    elif cat in ("kernel", "gpu_memcpy", "gpu_memset"):
        kernels_by_corr[e.get("args", {}).get("correlation")] = e
for k in by_tid_py:
    by_tid_py[k].sort(key=lambda e: e["ts"])
ops_by_tid = collections.defaultdict(list)
for e in ops:
    ops_by_tid[(e["pid"], e["tid"])].append(e)
for k in ops_by_tid:
    ops_by_tid[k].sort(key=lambda e: e["ts"])


starts = {k: [e["ts"] for e in v] for k, v in by_tid_py.items()}
op_starts = {k: [e["ts"] for e in v] for k, v in ops_by_tid.items()}


def innermost(lst, st, ts, pred, look=4000):
    i = bisect.bisect_right(st, ts) - 1
    n = 0
    while i >= 0 and n < look:
        e = lst[i]
        if e["ts"] + e.get("dur", 0) >= ts and pred(e):
            return e
        i -= 1; n += 1
    return None


PKG = ("xva-trainer_amd/", "xva_trainer_amd/")
is_pkg = lambda e: any(p in e["name"] for p in PKG)
sites = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
total = collections.Counter()
for r in runtime:
    name = r["name"]
    if not (name.startswith("hipMemcpy") or name.startswith("hipMemset") or name.startswith("hipLaunchKernel") or name.startswith("hipModuleLaunchKernel") or name.startswith("hipExtModuleLaunchKernel")):
        continue
    key = (r["pid"], r["tid"])
    dev = kernels_by_corr.get(r.get("args", {}).get("correlation"))
    kname = dev["name"] if dev else ""
    native = name.startswith("hipMemcpy") or name.startswith("hipMemset") or "at::native" in kname or "rocclr" in kname
    total["all"] += 1
    if not native:
        continue
    total["native"] += 1
    label = "memcpy" if name.startswith("hipMemcpy") else ("memset" if name.startswith("hipMemset") else kname.split("<")[0].replace("void at::native::", "")[:28] + ":" + (kname.split("at::native::")[2][:24] if kname.count("at::native::") > 1 else ""))
    aop = innermost(ops_by_tid.get(key, []), op_starts.get(key, []), r["ts"], lambda e: True, 50)
    aname = aop["name"] if aop else "?"
    pf = innermost(by_tid_py.get(key, []), starts.get(key, []), r["ts"], is_pkg)
    site = pf["name"].split("xva-trainer_amd/")[-1][:80] if pf else "(no package frame: %s)" % aname
    sites[site] += 1; kinds[site][label if pf is None else aname + "/" + label.split(":")[0]] += 1
print("runtime launches in the traced iteration: %d, torch-native (memcpy / memset / at::native kernels): %d" % (total["all"], total["native"]))
for s, n in sites.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 90):
    print("  %4d  %-82s %s" % (n, s, dict(kinds[s].most_common(4))))
