"""cProfile of ONE xVAPitch C5 iteration's host side (the iteration is issue-bound: ~3 600 launches through torch autograd glue + ctypes): top
functions by own time and by cumulative time.  python tools/c5_host_profile.py"""
import cProfile, io, os, pstats, runpy, sys
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    g["iteration"]()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
    print("\n".join(l[:190] for l in s.getvalue().splitlines()[:70]))
