"""A/B of the fp32-storage products (xva_gemm_set_fp32_products: 0 exact fp32 MFMA, 1 three bf16 MFMAs on hi + lo split operands) on the FastPitch
step of the headline configuration in the fp32 storage mode: python tools/fp32_products_ab.py"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xva_trainer_amd import _lib

a = argparse.Namespace(dropout=0.1, batch=32, t_text=150, t_mel=860, stage=3, no_roofline=True)
for mode in (0, 1):
    _lib.lib.xva_gemm_set_fp32_products(mode)
    r = bench.fastpitch_fp32_leg(a, "cuda:0", steps=5, warm=2)
    print("fp32 products mode %d: %.2f ms / step, %.0f mel-frames/s, loss %.6f" % (mode, r["ms_per_step"], r["value"], r["final_loss"]), flush=True)
_lib.lib.xva_gemm_set_fp32_products(0)
