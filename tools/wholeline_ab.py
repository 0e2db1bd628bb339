import sys, runpy
sys.path.insert(0, '/root/repo')
from xva_trainer_amd import _lib
mode = int(sys.argv[1]); tool = sys.argv[2]
_lib.lib.xva_gemm_set_wholeline(mode)
sys.argv = [tool] + sys.argv[3:]
runpy.run_path(tool, run_name="__main__")
